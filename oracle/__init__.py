"""CPU oracle package: test infrastructure only (see oracle/rk_oracle.py).

The product package `extensisq_amd` never imports anything from here.
"""
