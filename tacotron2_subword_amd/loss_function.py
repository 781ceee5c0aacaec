"""Tacotron2Loss (loss_function.py:6-66): MSE(mel) + MSE(mel_postnet) + BCE-with-logits(gate), means over all
elements including padding, plus the alignment-guide branches (alignloss = "L2" / "KL", active for iters < 40000).
Host-side torch reductions on the decoder's outputs; their gradients on the alignments
enter the HIP decoder backward through its d_align inputs.

The reference's quirks are kept because callers can observe them:
  * L2 compares BOTH alignments with the phone-level target, so it needs T_sub == T_in (nn.MSELoss raises otherwise);
  * KL slices `a[b][:mel_len[b]-1][:text_len[b]-1]` — both slices cut the FRAME axis — and takes mel_len from x[4],
    which is the (max_input_len, max_output_len) pair of parse_batch, so it only works for B <= 2 (IndexError beyond),
    uses max_input_len for item 0 and max_output_len for item 1;
  * KL replaces exact zeros by 1e-6.  The reference does that in place on the model outputs and on the target; here
    the edit is made on copies (same loss, same gradients) because the decoder's alignment buffer doubles as the
    saved recurrent state of the HIP backward."""
import torch
from torch import nn
from torch.nn import functional as F


class Tacotron2Loss(nn.Module):
    def __init__(self, alignloss=""):
        super().__init__()
        self.alignloss = alignloss

    def forward(self, model_output, targets, x=None, iters=0):
        mel_target, gate_target = targets[0], targets[1]
        mel_out, mel_out_postnet, gate_out = model_output[0], model_output[1], model_output[2]
        mel_loss = F.mse_loss(mel_out, mel_target) + F.mse_loss(mel_out_postnet, mel_target)
        gate_loss = F.binary_cross_entropy_with_logits(gate_out.reshape(-1, 1), gate_target.reshape(-1, 1))
        align_loss = align_bert_loss = None
        if self.alignloss == "L2" and iters < 40000:
            align_out, align_bert_out, align_target = model_output[3], model_output[4], targets[2]
            align_loss = nn.MSELoss()(align_out, align_target)
            align_bert_loss = nn.MSELoss()(align_bert_out, align_target)
        elif self.alignloss == "KL" and iters < 40000:
            align_out, align_bert_out, align_target = model_output[3], model_output[4], targets[2]
            eps = 0.000001
            align_out = align_out.masked_fill(align_out == 0, eps)
            align_bert_out = align_bert_out.masked_fill(align_bert_out == 0, eps)
            align_target = align_target.masked_fill(align_target == 0, eps)
            text_len, mel_len = x[1], x[4]
            align_loss = align_bert_loss = 0
            for b in range(align_target.size(0)):
                n_mel, n_text = int(mel_len[b]) - 1, int(text_len[b]) - 1
                aliout = align_out[b][:n_mel][:n_text]
                alibertout = align_bert_out[b][:n_mel][:n_text]
                alitar = align_target[b][:n_mel][:n_text]
                align_loss = align_loss + torch.mean(torch.sum(alitar * (torch.log(alitar) - torch.log(aliout)), dim=-1))
                align_bert_loss = align_bert_loss + torch.mean(torch.sum(alitar * (torch.log(alitar) - torch.log(alibertout)), dim=-1))
        total = mel_loss + gate_loss
        if align_loss is not None:
            total = total + align_loss + align_bert_loss
        return total, mel_loss, gate_loss, align_loss, align_bert_loss
