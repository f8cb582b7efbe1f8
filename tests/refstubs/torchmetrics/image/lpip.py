"""LPIPS needs AlexNet weights that are not in this image: the stand-in reports NaN so that the number the reference prints at
[REF mp_Mapper.py:422] reads as "not measured" instead of as a result."""
import torch


class LearnedPerceptualImagePatchSimilarity(torch.nn.Module):
    def __init__(self, net_type="alex", normalize=True, **kw):
        super().__init__()

    def forward(self, a, b):
        return torch.full((), float("nan"), device=a.device)
