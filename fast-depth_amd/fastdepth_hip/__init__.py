"""Host-side binding of libfastdepth_hip.so (C ABI: include/fastdepth_hip.h) for the FastDepth hot path.

``capi``   -- ctypes declarations of the C ABI (no torch).
``plan``   -- nn.Module tree -> fused-layer description (shape discovery from the live sub-modules, which
              is how pruned / unpickled reference checkpoints arrive; main.py:49-57).
``engine`` -- torch plumbing: device memory, current stream, weight-version tracking.
"""
