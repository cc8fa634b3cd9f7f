"""CPU oracle (test infrastructure only; see paradis_oracle.py header)."""
