"""clip-fsar_amd: MI355X-native CLIP-FSAR episodic-inference hot path.

Hand-written HIP (gfx950) kernels behind a C-ABI shared library (``csrc/`` ->
``libclipfsar_hip.so``, declared in ``include/clipfsar_hip.h``) and the host-side
Python mirror of the reference's ``models/base`` builder / registry surface.
"""
__version__ = "0.1.0"
