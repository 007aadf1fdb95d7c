"""Sequence models (mirrors spotlight/sequence): ImplicitSequenceModel with the PoolNet
representation runs on the fused gfx950 kernels of csrc/slk_seq.hip."""
