"""hmmufotu_amd — MI355X-native per-read assignment engine for HmmUFOtu databases.

The compute path lives in csrc/ (HIP kernels + C ABI, see include/hmmufotu_amd.h);
`engine` is the ctypes host binding, `synth` builds synthetic databases and reads.
"""
__version__ = "0.1.0"
