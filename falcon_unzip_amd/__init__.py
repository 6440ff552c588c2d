"""falcon_unzip_amd -- MI355X-native phasing engine behind FALCON_unzip's per-contig task API.

Only the hot path lives here: `phasing` / `phasing_readmap` mirror the reference
modules of the same name; `_lib` binds libfzphase.so (HIP kernels, include/fzphase.h); `sim`
generates synthetic inputs for tests and benchmarks.
"""
__version__ = "0.1.0"
