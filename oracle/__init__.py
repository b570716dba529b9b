"""CPU oracle for the Pix2Pix GAN training step of Affi-Amine/GAN-AUG-PFA.

TEST INFRASTRUCTURE ONLY.  This package is a numpy restatement of the
arithmetic the reference's hot path performs (``train_gan.py:46-75`` calling
``models.py:149-247``); the arithmetic itself lives in the reference's
third-party dependency PyTorch (``requirements.txt:1``, unpinned; the oracle was
pinned against torch 2.10.0+rocm7.0 CPU / oneDNN in the build container).

Pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md F6), so the pins are outputs of the reference itself, imported from
``/root/reference`` by ``tests/make_golden.py`` and committed as plain arrays
under ``tests/golden/``.  ``tests/test_oracle_golden.py`` checks every function
here against those vectors.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker / the reported CPU
baseline.  Nothing under ``gan-aug-pfa_amd/`` imports it: the product path runs
hand-written HIP through ``libpfa_hip.so`` and raises if that library is absent.
"""
