"""CPU oracle (test infrastructure only — see grapes_oracle.py header)."""
