"""CPU oracle for the archNEMESIS hot path -- TEST INFRASTRUCTURE ONLY (see ansfm_oracle.c)."""
