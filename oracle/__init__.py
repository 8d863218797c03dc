"""CPU oracle for the TimeAutoDiff hot path (TEST INFRASTRUCTURE ONLY).

This package is a numpy restatement of the reference's algorithm for the
denoiser training step, the reverse-diffusion sampler and the mixed-type VAE
forward/backward.  It is the checker the HIP path is compared against: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it.  Nothing under ``timeautodiff_amd/`` imports it and the product
path never falls back to it.

Pinning: every function here is checked against golden vectors produced by
importing the reference implementation itself (``tests/golden/make_fixtures.py``
writes them; ``tests/test_oracle_golden.py`` checks them).

Each function cites the reference ``file:line`` it restates (paths relative to
the reference checkout).
"""
from . import ops, rnn, denoiser, vae, optim, loops  # noqa: F401
