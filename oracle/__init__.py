"""CPU oracle (test infrastructure only -- see glm_oracle.py header)."""
