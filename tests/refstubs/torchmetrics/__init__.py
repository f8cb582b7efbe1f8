"""Stand-in for torchmetrics (only `torchmetrics.image.lpip` is imported [REF mp_Mapper.py:19])."""
