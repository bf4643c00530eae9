"""tinyedm_amd -- MI355X-native EDM/EDM2 training + sampling hot path (HIP kernels behind the
tinyedm Python API)."""
__version__ = "0.1.0"
