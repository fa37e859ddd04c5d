"""CPU oracle for the Silent-Speech per-clip hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``silent_speech_amd/`` imports this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may.  The product path runs on the HIP extension and fails
loudly when it is missing -- it never falls back to this code.

Parity status: PINNED.  ``tests/golden/*.npz`` were generated in the authoring
container by importing the reference's own ``train_model_official.py`` (and the
numeric helpers of ``record_landmarks_official.py`` / ``live_infer_official.py``)
with ``tests/golden/make_golden.py``; ``tests/test_oracle_golden.py`` checks
this restatement against those vectors.  The grayscale conversion / resize of
the crop (OpenCV, absent here) is NOT part of the oracle: unpinned and out of
the round-1 scope.
"""
