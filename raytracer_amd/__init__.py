"""MI355X-native ray-casting hot path (traverse -> sort -> shade) behind the C ABI of include/rt_hip.h.

api      ctypes binding of librt_hip.so and a mirror of the reference's RayTracer object
scene    scene construction through the C++ host library librt_host.so
tiling   row-band partition of a frame over ranks and the framebuffer gather
layout   numpy views of the C structures
"""
__all__ = ["api", "scene", "tiling", "layout"]
