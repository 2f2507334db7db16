"""TEST INFRASTRUCTURE ONLY -- synthetic checkpoints for the oracle side.

The generator itself is plain numpy (no hot-path arithmetic) and lives in
``masr_amd/utils/synthetic.py`` so that ``bench.py`` can build random-init weights without
importing the oracle; it is re-exported here for the tests and ``make_golden.py``.
"""
from masr_amd.utils.synthetic import (conformer_state_dict, deepspeech2_state_dict,  # noqa: F401
                                      efficient_conformer_state_dict,
                                      squeezeformer_state_dict, synthetic_pcm, synthetic_vocab)
