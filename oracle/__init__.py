"""CPU oracle for the flow2d hot path -- test infrastructure only (see flow2d_oracle.c)."""
