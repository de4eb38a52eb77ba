"""Synthetic workloads (configs, seeded weights, pixels, prompts) and a builder for the product model.  Not the oracle, not the product."""
