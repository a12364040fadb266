// model.cpp -- PLY ingest and wide bounding-box hierarchy build on the host.
//
// Restates what the reference does in model.cpp:25-222 (TriangleBlock::subdivide/reserve/fill,
// Model::load/prepare/put) on index arrays instead of pointer trees, so that the emitted groups,
// boxes, leaf-local vertices and packed triangles are byte-identical to the reference's for the same
// mesh.  Build with -ffp-contract=off: every float expression here is evaluated as written.

#include "model.h"
#include "parallel.h"

#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace rt {

// ---------------------------------------------------------------- ResourceManager

void ResourceManager::alloc()
{
    assert(!allocated_);
    groups_.assign(want_[0], rt_group());
    aabbs_.assign(want_[1], rt_aabb());
    vertices_.assign(want_[2], rt_vertex());
    triangles_.assign(want_[3], 0u);
    std::memset(groups_.data(), 0, groups_.size() * sizeof(rt_group));
    std::memset(aabbs_.data(), 0, aabbs_.size() * sizeof(rt_aabb));
    std::memset(vertices_.data(), 0, vertices_.size() * sizeof(rt_vertex));
    allocated_ = true;
}

bool ResourceManager::full() const
{
    for (int i = 0; i < 4; i++)
        if (used_[i] != want_[i]) return false;
    return true;
}

uint32_t ResourceManager::take(int which, size_t n)
{
    if (n && (!allocated_ || used_[which] + n > want_[which]))
        throw std::logic_error("ResourceManager: request exceeds the reserved size");
    const uint32_t first = uint32_t(used_[which]);
    used_[which] += n;
    return first;
}

// ---------------------------------------------------------------- camera

void set_camera(rt_camera &cam, size_t width, size_t height, float tan_fov, const Vec3 &pos, const Vec3 &view,
                const Vec3 &up)
{
    const float w = float(width), h = float(height);
    const float unit = tan_fov / std::sqrt(w * w + h * h);
    const Vec3 dir = normalized(view);
    const Vec3 dx = scale(normalized(cross(view, up)), unit);
    const Vec3 dy = cross(dx, dir);
    const Vec3 corner = dir - divide(scale(dx, w) + scale(dy, h), 2);
    store3(cam.eye, pos);
    store3(cam.top_left, corner);
    store3(cam.dx, dx);
    store3(cam.dy, dy);
    cam.width = uint32_t(width);
    cam.height = uint32_t(height);
}

// ---------------------------------------------------------------- PLY ingest

namespace {

// Whitespace-separated token reader over a file held in memory.
class Tokens {
public:
    explicit Tokens(std::vector<char> &&text) : text_(std::move(text)), at_(text_.data()) {}

    bool word(std::string &out)
    {
        skip_space();
        if (!*at_) return false;
        const char *start = at_;
        while (*at_ && !is_space(*at_)) at_++;
        out.assign(start, at_);
        return true;
    }

    bool expect(const char *literal)
    {
        std::string w;
        return word(w) && w == literal;
    }

    // The text of the comment line as the reference's template takes it ("comment %*[^\n] ", model.cpp:148): white
    // space of any kind first, line ends included, then at least one character, up to the end of that line.  (After
    // a bare "comment" the NEXT line is taken for the text, and the file then fails to load there as it does here.)
    bool comment_text()
    {
        skip_space();
        if (!*at_) return false;
        while (*at_ && *at_ != '\n') at_++;
        return true;
    }

    bool number(float &out)
    {
        skip_space();
        char *end;
        out = std::strtof(at_, &end);
        if (end == at_) return false;
        at_ = end;
        return true;
    }

    bool number(size_t &out)
    {
        skip_space();
        if (*at_ < '0' || *at_ > '9') return false;
        char *end;
        out = std::strtoull(at_, &end, 10);
        at_ = end;
        return true;
    }

    const char *mark() const { return at_; }
    const char *end() const { return text_.data() + text_.size() - 1; }  // the terminating NUL
    void rewind(const char *to) { at_ = to; }

private:
    static bool is_space(char c) { return c == ' ' || c == '\n' || c == '\r' || c == '\t' || c == '\f' || c == '\v'; }
    void skip_space()
    {
        while (is_space(*at_)) at_++;
    }

    std::vector<char> text_;
    const char *at_;
};

bool slurp(const char *file, std::vector<char> &text)
{
    FILE *in = std::fopen(file, "rb");
    if (!in) return false;
    std::fseek(in, 0, SEEK_END);
    const long size = std::ftell(in);
    std::fseek(in, 0, SEEK_SET);
    if (size < 0) {
        std::fclose(in);
        return false;
    }
    text.resize(size_t(size) + 1);
    const size_t got = std::fread(text.data(), 1, size_t(size), in);
    std::fclose(in);
    text[got] = '\0';
    return got == size_t(size);
}

// Binary cache of a parsed PLY file: vertex positions and face indices, keyed by a hash of the file's
// bytes.  Parsing the ASCII text of a large mesh with strtof is the slow part of start-up (the
// reference's fscanf loop, model.cpp:162-190, is slower still); reading the file and hashing it is
// not.  The cache lives in $RT_MESH_CACHE, or in rt_mesh_cache_<uid> under $TMPDIR or /tmp; an empty
// RT_MESH_CACHE switches it off.  A cache file that does not match in every header field, or whose size is not
// exactly what its counts say, is ignored.  The directory is only used when it belongs to this user and nobody
// else may write to it (a directory another local user created under /tmp could hold planted files).
struct CacheHeader {
    char magic[8];  // "RTMESH4": version of this layout and of the parser that filled it
    uint64_t hash, file_size, n_vtx, n_tri;
};

// 64-bit hash of the file's bytes, the cache's key: four interleaved multiply-xorshift lanes over 32-byte blocks
// (a byte-at-a-time FNV-1a took 70 ms for a 32 MB file, this takes 10), the tail and the length folded in at the end.
// A key for finding one's own earlier parse again, not a defence against forged files (see cache_path for that).
uint64_t content_hash(const char *data, size_t size)
{
    uint64_t lane[4] = {0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull, 0x27D4EB2F165667C5ull};
    size_t i = 0;
    for (; i + 32 <= size; i += 32) {
        uint64_t w[4];
        std::memcpy(w, data + i, 32);
        for (int k = 0; k < 4; k++) {
            lane[k] = (lane[k] ^ w[k]) * 0xFF51AFD7ED558CCDull;
            lane[k] ^= lane[k] >> 29;
        }
    }
    uint64_t h = size;
    for (; i < size; i++) h = (h ^ (unsigned char)data[i]) * 1099511628211ull;
    for (int k = 0; k < 4; k++) {
        h = (h ^ lane[k]) * 0xC4CEB9FE1A85EC53ull;
        h ^= h >> 32;
    }
    return h;
}

std::string cache_path(const char *file, uint64_t hash)
{
    const char *dir = std::getenv("RT_MESH_CACHE");
    std::string base;
    if (dir) {
        if (!*dir) return "";
        base = dir;
    } else {
        const char *tmp = std::getenv("TMPDIR");
        base = std::string(tmp && *tmp ? tmp : "/tmp") + "/rt_mesh_cache_" + std::to_string((unsigned long)getuid());
    }
    ::mkdir(base.c_str(), 0700);
    struct stat st;
    if (::lstat(base.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != getuid() || (st.st_mode & (S_IWGRP | S_IWOTH))) return "";
    std::string name = file;
    const size_t slash = name.find_last_of('/');
    if (slash != std::string::npos) name = name.substr(slash + 1);
    char hex[20];
    std::snprintf(hex, sizeof(hex), "%016llx", (unsigned long long)hash);
    return base + "/" + name + "." + hex + ".rtmesh";
}

bool read_cache(const std::string &path, uint64_t hash, uint64_t file_size, std::vector<Vec3> &positions,
                std::vector<std::array<uint32_t, 3>> &faces)
{
    FILE *in = std::fopen(path.c_str(), "rb");
    if (!in) return false;
    CacheHeader h;
    struct stat st;
    bool ok = ::fstat(fileno(in), &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == getuid() &&
              std::fread(&h, sizeof(h), 1, in) == 1 && !std::memcmp(h.magic, "RTMESH4", 8) && h.hash == hash &&
              h.file_size == file_size && h.n_vtx && h.n_tri && h.n_vtx < (1ull << 32) && h.n_tri < (1ull << 32);
    // the counts must account for the file's size exactly before anything is sized by them
    ok = ok && uint64_t(st.st_size) == sizeof(h) + h.n_vtx * sizeof(Vec3) + h.n_tri * sizeof(faces[0]);
    if (ok) {
        try {
            positions.resize(h.n_vtx);
            faces.resize(h.n_tri);
        } catch (const std::bad_alloc &) {
            ok = false;
        }
    }
    if (ok)
        ok = std::fread(positions.data(), sizeof(Vec3), h.n_vtx, in) == h.n_vtx &&
             std::fread(faces.data(), sizeof(faces[0]), h.n_tri, in) == h.n_tri && std::fgetc(in) == EOF;
    std::fclose(in);
    return ok;
}

void write_cache(const std::string &path, uint64_t hash, uint64_t file_size, const std::vector<Vec3> &positions,
                 const std::vector<std::array<uint32_t, 3>> &faces)
{
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    FILE *out = std::fopen(tmp.c_str(), "wb");
    if (!out) return;
    CacheHeader h;
    std::memcpy(h.magic, "RTMESH4", 8);
    h.hash = hash;
    h.file_size = file_size;
    h.n_vtx = positions.size();
    h.n_tri = faces.size();
    bool ok = std::fwrite(&h, sizeof(h), 1, out) == 1 && std::fwrite(positions.data(), sizeof(Vec3), positions.size(), out) == positions.size() &&
              std::fwrite(faces.data(), sizeof(faces[0]), faces.size(), out) == faces.size();
    ok = std::fclose(out) == 0 && ok;
    if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) std::remove(tmp.c_str());
}

}  // namespace

// The vertex and face lines of a large file on several threads.  Applies only when every element sits on a line of
// its own with exactly the tokens it needs (what every PLY writer emits): each thread then reads whole lines with the
// same strtof / strtoull calls the tokenizer makes, so the arrays are the tokenizer's.  Anything else -- too few lines,
// a line with a token missing or to spare, a face that is not "3 a b c" with indices in range -- returns false and leaves
// the decision to the tokenizer, which reads the file as the reference's fscanf does: as one stream of tokens.
namespace {

inline bool line_space(char c) { return c == ' ' || c == '\r' || c == '\t' || c == '\f' || c == '\v'; }

size_t parallel_parse_min_bytes()
{
    if (const char *env = std::getenv("RT_PLY_PARALLEL_MIN")) return size_t(std::strtoull(env, nullptr, 10));
    return size_t(1) << 20;
}

bool parse_lines(const char *begin, const char *end, size_t n_vtx, size_t n_tri, size_t extra, std::vector<Vec3> &positions,
                 std::vector<std::array<uint32_t, 3>> &faces)
{
    const size_t bytes = size_t(end - begin), threads = host_threads();
    if (threads < 2 || bytes < parallel_parse_min_bytes()) return false;
    const size_t chunks = std::min<size_t>(4 * threads, bytes / 4096 + 1);
    std::vector<const char *> start(chunks + 1, end);
    start[0] = begin;
    for (size_t c = 1; c < chunks; c++) {  // every chunk starts at the start of a line
        const char *at = begin + bytes / chunks * c;
        while (at < end && at[-1] != '\n') at++;
        start[c] = at;
    }
    auto next_line = [](const char *at, const char *stop) {  // first character of the next non-empty line, or stop
        while (at < stop && (line_space(*at) || *at == '\n')) at++;
        return at;
    };
    std::vector<size_t> lines(chunks + 1, 0);
    parallel_for(chunks, [&](size_t c) {
        size_t n = 0;
        for (const char *at = next_line(start[c], start[c + 1]); at < start[c + 1]; n++) {
            while (at < start[c + 1] && *at != '\n') at++;
            at = next_line(at, start[c + 1]);
        }
        lines[c + 1] = n;
    });
    for (size_t c = 0; c < chunks; c++) lines[c + 1] += lines[c];
    if (lines[chunks] < n_vtx + n_tri) return false;
    positions.assign(n_vtx, Vec3());
    faces.assign(n_tri, std::array<uint32_t, 3>());
    std::atomic<bool> ok(true);
    parallel_for(chunks, [&](size_t c) {
        size_t line = lines[c];
        const char *stop = start[c + 1];
        for (const char *at = next_line(start[c], stop); at < stop && line < n_vtx + n_tri && ok.load(std::memory_order_relaxed); line++) {
            auto token = [&]() {  // at the next token of this line, or false at the line's end
                while (at < stop && line_space(*at)) at++;
                return at < stop && *at != '\n';
            };
            bool good = true;
            if (line < n_vtx) {
                float v[3 + 4] = {0, 0, 0, 0, 0, 0, 0};
                for (size_t k = 0; k < 3 + extra && good; k++) {
                    char *after;
                    good = token() && (v[k] = std::strtof(at, &after), after != at);
                    if (good) at = after;
                }
                positions[line] = Vec3(v[0], v[1], v[2]);
            } else {
                size_t v[4] = {0, 0, 0, 0};
                for (int k = 0; k < 4 && good; k++) {
                    char *after;
                    good = token() && *at >= '0' && *at <= '9' && (v[k] = std::strtoull(at, &after, 10), true);
                    if (good) at = after;
                }
                good = good && v[0] == 3 && v[1] < n_vtx && v[2] < n_vtx && v[3] < n_vtx;
                if (good) faces[line - n_vtx] = {uint32_t(v[1]), uint32_t(v[2]), uint32_t(v[3])};
            }
            if (!good || token()) {  // a token missing, malformed or to spare
                ok.store(false, std::memory_order_relaxed);
                return;
            }
            at = next_line(at, stop);
        }
    });
    return ok.load();
}

}  // namespace

bool Model::load(const char *file)
{
    try {
        return load_file(file);
    } catch (const std::bad_alloc &) {  // a header that promises more than memory holds: a failed load, not a crash
        pos_.clear();
        faces_.clear();
        return false;
    }
}

bool Model::load_file(const char *file)
{
    if (!pos_.empty() || !faces_.empty()) return false;
    std::vector<char> text;
    if (!slurp(file, text)) return false;
    const uint64_t file_size = text.size() - 1, hash = content_hash(text.data(), file_size);
    const std::string cached = cache_path(file, hash);
    if (!cached.empty()) {
        std::vector<Vec3> positions;
        std::vector<std::array<uint32_t, 3>> faces;
        if (read_cache(cached, hash, file_size, positions, faces) && assign(positions, faces)) {
            from_cache_ = true;
            return true;
        }
        pos_.clear();
        faces_.clear();
    }
    Tokens tok(std::move(text));

    // Header, in the fixed order the reference's scanf templates demand (model.cpp:145-160).
    size_t n_vtx = 0, n_tri = 0;
    if (!tok.expect("ply") || !tok.expect("format") || !tok.expect("ascii") || !tok.expect("1.0")) return false;
    if (!tok.expect("comment") || !tok.comment_text()) return false;
    if (!tok.expect("element") || !tok.expect("vertex") || !tok.number(n_vtx) || !n_vtx) return false;
    for (const char *axis : {"x", "y", "z"})
        if (!tok.expect("property") || !tok.expect("float") || !tok.expect(axis)) return false;
    size_t extra = 0;  // further float properties per vertex, skipped when reading
    for (;;) {
        const char *here = tok.mark();
        std::string name;
        if (tok.expect("property") && tok.expect("float") && tok.word(name)) {
            extra++;
            continue;
        }
        tok.rewind(here);
        break;
    }
    if (extra > 4) return false;  // the reference's format buffer holds four (model.cpp:143, 166-171)
    if (!tok.expect("element") || !tok.expect("face") || !tok.number(n_tri) || !n_tri) return false;
    std::string list_type;
    if (!tok.expect("property") || !tok.expect("list") || !tok.expect("uchar") || !tok.word(list_type)) return false;
    if (list_type != "int" && list_type != "uint") return false;
    if (!tok.expect("vertex_indices") || !tok.expect("end_header")) return false;

    std::vector<Vec3> positions;
    std::vector<std::array<uint32_t, 3>> faces;
    if (!parse_lines(tok.mark(), tok.end(), n_vtx, n_tri, extra, positions, faces)) {
        positions.assign(n_vtx, Vec3());
        for (size_t i = 0; i < n_vtx; i++) {
            float skipped;
            if (!tok.number(positions[i].x) || !tok.number(positions[i].y) || !tok.number(positions[i].z)) return false;
            for (size_t k = 0; k < extra; k++)
                if (!tok.number(skipped)) return false;
        }
        faces.assign(n_tri, std::array<uint32_t, 3>());
        for (size_t i = 0; i < n_tri; i++) {
            size_t corners, v[3];
            if (!tok.number(corners) || corners != 3) return false;
            for (int k = 0; k < 3; k++) {
                if (!tok.number(v[k]) || v[k] >= n_vtx) return false;
                faces[i][k] = uint32_t(v[k]);
            }
        }
    }
    if (!assign(positions, faces)) return false;
    if (!cached.empty()) write_cache(cached, hash, file_size, positions, faces);
    return true;
}

bool Model::assign(const std::vector<Vec3> &positions, const std::vector<std::array<uint32_t, 3>> &faces)
{
    if (!pos_.empty() || !faces_.empty() || positions.empty() || faces.empty()) return false;
    for (const auto &f : faces)
        for (uint32_t v : f)
            if (v >= positions.size()) return false;
    pos_ = positions;
    faces_.resize(faces.size());
    for (size_t i = 0; i < faces.size(); i++)
        for (int k = 0; k < 3; k++) faces_[i].v[k] = faces[i][k];
    prepare();
    return true;
}

// Area-weighted vertex normals, triangle centroids and the centroid bounds of the root (model.cpp:197-214).
void Model::prepare()
{
    norm_.assign(pos_.size(), Vec3(0, 0, 0));
    slot_.assign(pos_.size(), -1);
    order_.resize(faces_.size());
    Bounds centroid_box;
    for (size_t i = 0; i < faces_.size(); i++) {
        Face &f = faces_[i];
        const Vec3 a = pos_[f.v[0]], b = pos_[f.v[1]], c = pos_[f.v[2]];
        f.center = divide((a + b) + c, 3);
        const Vec3 n = cross(b - a, c - a);
        for (int k = 0; k < 3; k++) norm_[f.v[k]] = norm_[f.v[k]] + n;
        centroid_box.add(f.center);
        order_[i] = uint32_t(i);
    }
    for (size_t i = 0; i < pos_.size(); i++) norm_[i] = divide(norm_[i], length(norm_[i]));
    nodes_.clear();
    Node root;
    root.first = 0;
    root.count = uint32_t(faces_.size());
    root.box = centroid_box;
    nodes_.push_back(root);
}

// ---------------------------------------------------------------- hierarchy

// The hierarchy is built level by level: the nodes of one level hold disjoint ranges of the triangle order, so
// their sorts run side by side (parallel_for), each exactly the sort the reference's recursion performs on that
// range (model.cpp:28-40) -- a node's range is left by its parent's sort in the same order either way.  Which
// nodes become box groups is then counted from the leaves up (model.cpp:42-46).
void Model::subdivide(size_t tri_threshold, size_t aabb_threshold)
{
    assert(nodes_.size() == 1);
    std::vector<int> level(1, 0), todo;
    while (!level.empty()) {
        todo.clear();
        for (int node : level)
            if (nodes_[node].count >= tri_threshold && nodes_[node].count >= 2) todo.push_back(node);  // (fewer than two cannot be halved)
        const size_t first_child = nodes_.size();
        nodes_.resize(first_child + 2 * todo.size());  // (no thread grows the vector)
        parallel_for(todo.size(), [&](size_t i) { split_node(todo[i], int(first_child + 2 * i)); });
        level.clear();
        for (size_t i = 0; i < 2 * todo.size(); i++) level.push_back(int(first_child + i));
    }
    count_boxes(0, aabb_threshold, true);
}

// Median split of one node along the longest axis of its centroid bounds (model.cpp:28-40).  The triangles are
// sorted as (centroid coordinate, triangle) pairs: std::sort makes the same comparisons in the same order as on
// the triangle numbers with an indirect comparison, so it leaves the same permutation -- ties included, which an
// unstable sort resolves by its own moves -- without a scattered read per comparison.
void Model::split_node(int node, int first_child)
{
    const uint32_t first = nodes_[node].first, count = nodes_[node].count;
    const Bounds box = nodes_[node].box;
    const Vec3 extent = box.hi - box.lo;
    int axis;
    if (extent.x > extent.y && extent.x > extent.z) axis = 0;
    else if (extent.y > extent.z) axis = 1;
    else axis = 2;

    struct Keyed {
        float key;
        uint32_t face;
    };
    uint32_t *begin = order_.data() + first;
    std::vector<Keyed> keyed(count);
    for (uint32_t i = 0; i < count; i++) keyed[i] = Keyed{faces_[begin[i]].center[axis], begin[i]};
    std::sort(keyed.begin(), keyed.end(), [](const Keyed &a, const Keyed &b) { return a.key < b.key; });
    for (uint32_t i = 0; i < count; i++) begin[i] = keyed[i].face;

    const uint32_t half = count / 2;
    Node lower, upper;
    lower.first = first;
    lower.count = half;
    lower.box = box;
    lower.box.hi[axis] = keyed[half - 1].key;
    upper.first = first + half;
    upper.count = count - half;
    upper.box = box;
    upper.box.lo[axis] = keyed[half].key;
    nodes_[first_child] = lower;
    nodes_[first_child + 1] = upper;
    nodes_[node].child[0] = first_child;
    nodes_[node].child[1] = first_child + 1;
}

// Returns how many boxes this subtree contributes to the closest enclosing box group: a node with at least
// aabb_threshold leaves below it (or the root) becomes a group of its own and counts as one (model.cpp:42-46).
size_t Model::count_boxes(int node, size_t aabb_threshold, bool is_root)
{
    if (nodes_[node].child[0] < 0) return 1;
    const size_t below = count_boxes(nodes_[node].child[0], aabb_threshold, false) + count_boxes(nodes_[node].child[1], aabb_threshold, false);
    if (!is_root && below < aabb_threshold) return below;
    nodes_[node].aabb_count = uint32_t(below);
    return 1;
}

void Model::reserve(ResourceManager &mngr)
{
    reserve_node(0, mngr);
}

void Model::reserve_node(int node, ResourceManager &mngr)
{
    Node &nd = nodes_[node];
    if (nd.child[0] >= 0) {
        reserve_node(nd.child[0], mngr);
        reserve_node(nd.child[1], mngr);
        if (nd.aabb_count) {
            mngr.reserve_groups(1);
            mngr.reserve_aabbs(nd.aabb_count);
        }
        return;
    }
    // Leaf: one mesh group, its triangles, and one copy of every vertex it touches (model.cpp:57-66).
    mngr.reserve_groups(1);
    mngr.reserve_triangles(nd.count);
    int used = 0;
    for (uint32_t i = 0; i < nd.count; i++)
        for (uint32_t v : faces_[order_[nd.first + i]].v)
            if (slot_[v] < 0) slot_[v] = used++;
    for (uint32_t i = 0; i < nd.count; i++)
        for (uint32_t v : faces_[order_[nd.first + i]].v) slot_[v] = -1;
    if (used >= (1 << 10)) throw std::length_error("leaf touches more than 1023 vertices");
    nd.vtx_count = uint32_t(used);
    mngr.reserve_vertices(nd.vtx_count);
}

void Model::fill(ResourceManager &mngr, uint32_t material_id)
{
    group_id_ = fill_node(0, mngr, material_id, nullptr);
}

// Emits the subtree in depth-first order: a box group takes its table slot and its box range before
// its children, a leaf takes slot, vertex range and triangle range when reached (model.cpp:76-136).
// aabb_cursor points at the next free box of the enclosing group, or is null at the model root.
uint32_t Model::fill_node(int node, ResourceManager &mngr, uint32_t material_id, uint32_t *aabb_cursor)
{
    const Node nd = nodes_[node];
    uint32_t id = 0;
    Bounds box;

    if (nd.child[0] >= 0 && nd.aabb_count) {
        const uint32_t slot = mngr.get_groups(1);
        const uint32_t first_box = mngr.get_aabbs(nd.aabb_count);
        rt_group *grp = mngr.group(slot);
        grp->aabb.aabb_offs = first_box;
        grp->aabb.aabb_count = nd.aabb_count;
        grp->aabb.flags = 0;
        uint32_t cursor = first_box;
        fill_node(nd.child[0], mngr, material_id, &cursor);
        fill_node(nd.child[1], mngr, material_id, &cursor);
        assert(cursor == first_box + nd.aabb_count);
        box.add(nodes_[nd.child[0]].box);
        box.add(nodes_[nd.child[1]].box);
        nodes_[node].box = box;
        id = make_group_id(slot, RT_TR_ORTHO, RT_SH_AABB);
    } else if (nd.child[0] >= 0) {
        // Too few leaves for a group of its own: the leaves join the enclosing group's box list.
        fill_node(nd.child[0], mngr, material_id, aabb_cursor);
        fill_node(nd.child[1], mngr, material_id, aabb_cursor);
        box.add(nodes_[nd.child[0]].box);
        box.add(nodes_[nd.child[1]].box);
        nodes_[node].box = box;
        return 0;
    } else {
        const uint32_t slot = mngr.get_groups(1);
        rt_group *grp = mngr.group(slot);
        grp->mesh.vtx_offs = mngr.get_vertices(nd.vtx_count);
        grp->mesh.tri_offs = mngr.get_triangles(nd.count);
        grp->mesh.tri_count = nd.count;
        grp->mesh.material_id = material_id;
        rt_vertex *vtx_out = nd.vtx_count ? mngr.vertex(grp->mesh.vtx_offs) : nullptr;
        uint32_t *tri_out = mngr.triangle(grp->mesh.tri_offs);
        int used = 0;
        for (uint32_t i = 0; i < nd.count; i++) {
            uint32_t packed = 0;
            int shift = 0;
            for (uint32_t v : faces_[order_[nd.first + i]].v) {
                if (slot_[v] < 0) {
                    slot_[v] = used++;
                    store3(vtx_out[slot_[v]].pos, pos_[v]);
                    store3(vtx_out[slot_[v]].norm, norm_[v]);
                    box.add(pos_[v]);
                }
                packed |= uint32_t(slot_[v]) << shift;
                shift += 10;
            }
            tri_out[i] = packed;
        }
        for (uint32_t i = 0; i < nd.count; i++)
            for (uint32_t v : faces_[order_[nd.first + i]].v) slot_[v] = -1;
        assert(uint32_t(used) == nd.vtx_count);
        nodes_[node].box = box;
        id = make_group_id(slot, RT_TR_ORTHO, RT_SH_MESH);
    }

    if (aabb_cursor) {
        rt_aabb *out = mngr.aabb((*aabb_cursor)++);
        store3(out->min, box.lo);
        store3(out->max, box.hi);
        out->group_id = id;
        out->local_id = 0;
    }
    return id;
}

void Model::put(rt_aabb &aabb, const rt_matrix &mat, uint32_t local_id) const
{
    assert(group_id_);
    Bounds box;
    for (const Vec3 &p : pos_) box.add(transform_point(mat, p));
    store3(aabb.min, box.lo);
    store3(aabb.max, box.hi);
    aabb.group_id = group_id_;
    aabb.local_id = local_id;
}

}  // namespace rt
