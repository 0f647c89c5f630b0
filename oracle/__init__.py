"""CPU oracle (test infrastructure only).  See laff_oracle.py."""
