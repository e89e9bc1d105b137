"""Placeholder module for the AST network_g class: the window-attention kernels are not built yet,
so `AST` is deliberately absent and `define_network({"type": "AST"})` raises the registry's
`ValueError("AST is not found.")` instead of silently running anything else."""
