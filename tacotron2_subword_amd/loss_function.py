"""Tacotron2Loss (loss_function.py:6-66), default branch: MSE(mel) + MSE(mel_postnet) +
BCE-with-logits(gate), means over all elements including padding.  The alignment-guide
branches (alignloss = "L2" / "KL") are SURVEY.md §8f N2 ('next')."""
from torch import nn
from torch.nn import functional as F


class Tacotron2Loss(nn.Module):
    def __init__(self, alignloss=""):
        super().__init__()
        if alignloss not in ("", None):
            raise NotImplementedError("alignment-guide losses (L2/KL) are not built yet (SURVEY.md §8f N2)")
        self.alignloss = alignloss

    def forward(self, model_output, targets, x=None, iters=0):
        mel_target, gate_target = targets[0], targets[1]
        mel_out, mel_out_postnet, gate_out = model_output[0], model_output[1], model_output[2]
        mel_loss = F.mse_loss(mel_out, mel_target) + F.mse_loss(mel_out_postnet, mel_target)
        gate_loss = F.binary_cross_entropy_with_logits(gate_out.reshape(-1, 1), gate_target.reshape(-1, 1))
        return mel_loss + gate_loss, mel_loss, gate_loss, None, None
