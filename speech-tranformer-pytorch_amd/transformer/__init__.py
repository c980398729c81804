"""Drop-in for the reference's ``transformer`` package (same import paths, class
names, constructor / forward signatures and ``state_dict`` keys), with the
arithmetic done by hand-written gfx950 HIP kernels (../csrc) through the C-ABI
library libst_hip.so.  Put this package's parent directory first on
``PYTHONPATH`` and ``train.py`` / ``train_multi.py`` import it unchanged.
"""
