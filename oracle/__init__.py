"""CPU oracle for the speech-transformer training step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / the timed CPU baseline.
The product path (``speech-tranformer-pytorch_amd/``) never imports this
package and fails loudly when its HIP library is missing.
"""
from .speech_transformer_oracle import *  # noqa: F401,F403
from . import feature_oracle  # noqa: F401  (CMVN / frame stacking / subsampling, Dataset.py)
from . import beam_oracle  # noqa: F401  (beam-search decode restatement; parity unpinned - see its header)
