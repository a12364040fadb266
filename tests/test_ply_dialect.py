"""Which PLY files load, and what they load as: raytracer_amd/host/model.cpp (an in-memory tokenizer) against the
REFERENCE's own loader (model.cpp:139-195, fscanf templates), compiled from /root/reference as
oracle/_ref/libref_model.so.  Every variant of a small mesh must be accepted by both with byte-identical scene arrays,
or refused by both."""
import os

import numpy as np
import pytest

from oracle import pyoracle as O
from raytracer_amd import layout as L, scene as S

pytestmark = pytest.mark.skipif(O.ref_model_lib() is None, reason="oracle/_ref/libref_model.so not built (needs /root/reference)")


def mesh(n=13):
    """A bumpy n x n height field: 2 (n-1)^2 triangles, enough for a few splits of the hierarchy."""
    ys, xs = np.mgrid[0:n, 0:n]
    z = 0.03 * np.sin(1.7 * xs) * np.cos(1.3 * ys)
    pos = np.stack([0.1 * xs / (n - 1) - 0.05, 0.1 * ys / (n - 1) + 0.05, z], axis=-1).reshape(-1, 3)
    faces = []
    for y in range(n - 1):
        for x in range(n - 1):
            a, b, c, d = y * n + x, y * n + x + 1, (y + 1) * n + x, (y + 1) * n + x + 1
            faces += [(a, b, c), (b, d, c)]
    return pos, faces


def ply_text(pos, faces, comment="comment made for a test\n", extras=(), list_type="int", vertex_fmt="%.7g", eol="\n",
             face_prefix="3 ", count_vtx=None, count_tri=None, tail=""):
    lines = ["ply", "format ascii 1.0"]
    text = eol.join(lines) + eol + comment.replace("\n", eol)
    text += f"element vertex {len(pos) if count_vtx is None else count_vtx}{eol}"
    for name in ("x", "y", "z") + tuple(extras):
        text += f"property float {name}{eol}"
    text += f"element face {len(faces) if count_tri is None else count_tri}{eol}property list uchar {list_type} vertex_indices{eol}end_header{eol}"
    for p in pos:
        text += " ".join(vertex_fmt % v for v in p) + "".join(" 0.5" for _ in extras) + eol
    for f in faces:
        text += face_prefix + " ".join(str(i) for i in f) + eol
    return text + tail


def both_loaders(path):
    """(ours, reference): scene arrays, or None where the loader refuses the file."""
    try:
        sc = S.build_scene([path], 64, 64, 8192, n_obj=2, tri_threshold=32, aabb_threshold=4)
        mine = (sc.aabbs, sc.vertices, sc.triangles, sc.matrices)
    except RuntimeError as e:
        assert "cannot load model" in str(e)
        mine, sc = None, None
    matrices = S.build_scene([S.BUNNY], 64, 64, 8192, n_obj=2).matrices if sc is None else sc.matrices
    try:
        _, aabbs, vertices, triangles = O.ref_model_build([path], matrices, 32, 4)
        ref = (aabbs, vertices, triangles)
    except RuntimeError:
        ref = None
    return mine, ref


POS, FACES = mesh()
VARIANTS = {
    # accepted by the reference
    "plain": dict(),
    "uint list": dict(list_type="uint"),
    "four more float properties": dict(extras=("nx", "ny", "nz", "confidence")),
    "one more float property": dict(extras=("confidence",)),
    "carriage returns": dict(eol="\r\n"),
    "exponent notation": dict(vertex_fmt="%.6e"),
    "trailing text": dict(tail="anything after the faces\n"),
    "blank lines in the comment's place": dict(comment="comment    spaced   out   \n"),
    # refused by the reference
    "no comment line": dict(comment=""),
    "two comment lines": dict(comment="comment one\ncomment two\n"),
    "empty comment": dict(comment="comment\n"),
    "five more float properties": dict(extras=("a", "b", "c", "d", "e")),
    "short list type": dict(list_type="short"),
    "no vertices": dict(count_vtx=0),
    "no faces": dict(count_tri=0),
    "more faces promised than given": dict(count_tri=len(FACES) + 1),
    "more vertices promised than given": dict(count_vtx=len(POS) + 1),
    "quads": dict(face_prefix="4 "),
}


def _reflow(text, per_line):
    """The same tokens with `per_line` of them on every data line: no longer one element per line."""
    head, body = text.split("end_header\n")
    tokens = body.split()
    lines = [" ".join(tokens[i:i + per_line]) for i in range(0, len(tokens), per_line)]
    return head + "end_header\n" + "\n".join(lines) + "\n"



@pytest.mark.parametrize("parser", ["tokenizer", "lines"])
@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_loader_agrees_with_the_reference(tmp_path, monkeypatch, name, parser):
    """parser: the tokenizer alone (one thread), or the line-by-line parser of large files first (forced here on a small
    one), which hands anything that is not one element per line back to the tokenizer."""
    monkeypatch.setenv("RT_MESH_CACHE", "")          # the text parser, not the binary cache
    monkeypatch.setenv("RT_HOST_THREADS", "1" if parser == "tokenizer" else "4")
    monkeypatch.setenv("RT_PLY_PARALLEL_MIN", "0")
    path = str(tmp_path / "mesh.ply")
    with open(path, "w", newline="") as f:
        f.write(ply_text(POS, FACES, **VARIANTS[name]))
    mine, ref = both_loaders(path)
    assert (mine is None) == (ref is None), f"{name}: ours {'refuses' if mine is None else 'accepts'}, the reference {'refuses' if ref is None else 'accepts'}"
    if mine is not None:
        aabbs, vertices, triangles, _ = mine
        assert aabbs.tobytes() == ref[0].tobytes(), name
        assert np.array_equal(vertices["pos"][:, :3].view(np.uint32), ref[1]["pos"][:, :3].view(np.uint32)), name
        assert np.array_equal(vertices["norm"][:, :3].view(np.uint32), ref[1]["norm"][:, :3].view(np.uint32)), name
        assert np.array_equal(triangles, ref[2]), name


def test_out_of_range_index_is_refused_by_both(tmp_path, monkeypatch):
    monkeypatch.setenv("RT_MESH_CACHE", "")
    faces = list(FACES)
    faces[5] = (0, 1, len(POS))
    path = str(tmp_path / "mesh.ply")
    open(path, "w").write(ply_text(POS, faces))
    assert both_loaders(path) == (None, None)
    faces[5] = (0, -1, 2)
    open(path, "w").write(ply_text(POS, faces))
    assert both_loaders(path) == (None, None)


@pytest.mark.parametrize("per_line", [1, 5, 7])
def test_elements_need_not_sit_on_lines_of_their_own(tmp_path, monkeypatch, per_line):
    """The reference reads a stream of tokens (fscanf), wherever the line ends fall: so does the loader, also when the
    line-by-line parser of large files gets the file first."""
    monkeypatch.setenv("RT_MESH_CACHE", "")
    monkeypatch.setenv("RT_HOST_THREADS", "4")
    monkeypatch.setenv("RT_PLY_PARALLEL_MIN", "0")
    path = str(tmp_path / "mesh.ply")
    open(path, "w").write(_reflow(ply_text(POS, FACES), per_line))
    mine, ref = both_loaders(path)
    assert mine is not None and ref is not None
    assert mine[0].tobytes() == ref[0].tobytes() and np.array_equal(mine[2], ref[2])
    assert np.array_equal(mine[1]["pos"][:, :3].view(np.uint32), ref[1]["pos"][:, :3].view(np.uint32))
