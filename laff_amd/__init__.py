"""laff_amd -- MI355X-native (gfx950) implementation of the LAFF retrieval hot path.

Python surface mirrors /root/reference (model.model.get_model, towers, predict, evaluation.eval, BigFile);
arithmetic runs in liblaff_hip.so (hand-written HIP kernels) through ctypes.  No CPU fallback.
"""
__version__ = '0.1.0'
