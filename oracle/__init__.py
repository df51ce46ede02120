"""oracle/ -- TEST INFRASTRUCTURE (CPU checker for the HIP path).  See cvvae_oracle.py header."""
