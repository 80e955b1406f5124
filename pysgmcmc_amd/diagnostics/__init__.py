"""Diagnostics: toy targets and cross-chain statistics on the GPU."""
