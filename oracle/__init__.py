"""CPU oracle for the spotlight hot path -- TEST INFRASTRUCTURE ONLY (see slk_oracle.c)."""
