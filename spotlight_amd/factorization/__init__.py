"""Factorization models (mirror spotlight/factorization)."""
