"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the MASR inference hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product (``masr_amd``) never imports
this package and fails loudly when its HIP library is missing.

Contents
--------
* ``fbank.py``      numpy restatement of ``torchaudio.compliance.kaldi.fbank``
                    (third-party, NOT vendored in the reference; call site
                    ``masr/data_utils/featurizer/audio_featurizer.py:120-138``)
                    plus the AudioSegment dB-normalise / int16 logic
                    (``masr/data_utils/audio.py:287-304,549-574``).
* ``conformer.py``  torch-CPU fp32 functional restatement of the reference
                    Conformer encoder + CTC head (``masr/model_utils/conformer``).
* ``decoders.py``   restatement of ``masr/decoders/ctc_greedy_decoder.py``.
* ``shims.py``      import shims that let the *unmodified* reference package be
                    imported from ``/root/reference`` in the build container.
* ``make_golden.py`` runs the real reference modules (through the shims) and
                    writes the fixtures under ``tests/golden/``.

Parity pinning: conformer / decoders / AudioSegment logic are pinned against
the reference's own code run in the build container (fixtures committed under
``tests/golden``).  fbank: **parity unpinned at the reference level** --
torchaudio is not installed here and the reference ships no fbank vectors; the
restatement is cross-checked against ``transformers.audio_utils`` (an
independent numpy Kaldi mimic) instead.
"""
