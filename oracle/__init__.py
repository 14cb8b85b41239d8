"""CPU oracle package (test infrastructure only; see pearson_oracle.py)."""
