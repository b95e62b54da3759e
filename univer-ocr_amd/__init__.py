"""univer-ocr MI355X backend: HIP kernels behind a C ABI (csrc/, include/univer_hip.h) and the
host-side mirror of the reference's nn framework API (nn/, my_model/)."""
__version__ = '0.2.0'
