"""st_amd - host side of the MI355X-native speech-transformer training step.

``native``      ctypes binding of libst_hip.so (the C-ABI HIP kernels)
``functional``  torch.autograd.Functions composed from the native kernels
``arena``       flat parameter / gradient / bf16-shadow storage
``dp``          RCCL data-parallel gradient all-reduce (the Horovod role)
``trainer``     the train.py step sequence

The drop-in nn.Modules live in the sibling package ``transformer`` (same import
paths as the reference).  There is no CPU fallback: without the HIP library and
a GPU the compute entry points raise.
"""
__version__ = "0.1.0"
