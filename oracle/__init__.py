"""CPU oracle for the JDC pitch-extractor training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pitchextractor_amd/`` may import
this package: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and there only as the checker.

Every function here is a plain numpy / stock-``torch`` CPU restatement of one
piece of the reference (martinambrus/PitchExtractor) and cites the reference
``file:line`` it follows.  Parity pinning:

* model / optimiser / trainer-step / collation / align_length / glide
  generator: pinned by golden vectors captured from the reference's own
  importable modules (``tests/golden/make_golden.py``).
* mel front end: the arithmetic lives in ``torchaudio`` (un-vendored,
  unpinned, not installed here) -> **parity unpinned by the reference**;
  pinned instead by two independent restatements that must agree
  (float64 direct DFT vs ``torch.stft``) -- see ``oracle/mel_ref.py``.
"""
