"""CPU oracle for the Emma-X hot path -- TEST INFRASTRUCTURE ONLY (see emmax_oracle.py header)."""
