"""CPU oracle for the ARCTE hot path -- TEST INFRASTRUCTURE ONLY (see oracle.py)."""
