"""oracle/ -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything from this package, and
only as the checker / reported baseline -- never as a product path (babyai_amd has no CPU path at all).

  shim/            restated gym / gym_minigrid / blosc (absent third-party dependencies of the reference;
                   PARITY UNPINNED against the real gym_minigrid, see shim/gym_minigrid/__init__.py)
  levels.py        stand-alone oracle: the reference's level layer (levelgen.py, verifier.py, all 105 level classes)
                   restated in plain Python; PINNED to the reference itself by the golden traces in tests/golden/
                   (tools/gen_golden.py) and, in the build container, by tests/test_oracle_reference.py
  refenv.py        import helper that runs /root/reference/babyai UNMODIFIED on the shim (build container only)
  cpu_baseline.py  bench.py's cpu_baseline leg (kind = "port")
"""
