"""TEST INFRASTRUCTURE — CPU oracle for the adversarial-evaluation hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this
package.  The shipped package ``audio_deepfake_adversarial_attacks_amd`` never does (tests/test_layout.py
greps for it).  Parity status: pinned against golden vectors generated from the reference itself
(tests/golden/generate_golden.py, run in the build container where /root/reference is importable).
"""
