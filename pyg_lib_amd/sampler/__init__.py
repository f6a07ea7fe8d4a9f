"""Placeholder until the HIP sampler lands."""
