"""CPU oracle for the S^3 hot path -- test infrastructure only (see oracle/s3_oracle.c)."""
