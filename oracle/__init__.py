"""TEST INFRASTRUCTURE ONLY — CPU restatement (oracle) of the ThermoNeRF rendering hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it, and only as the checker / the timed CPU baseline.
The product package (``thermo_nerf_amd``) never imports this package.
"""
