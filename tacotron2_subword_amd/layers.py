"""LinearNorm / ConvNorm parameter containers (layers.py:8-39): nn.Linear / nn.Conv1d with
Xavier-uniform init; the parameters live at ``.linear_layer.*`` / ``.conv.*`` (state_dict contract)."""
import torch
from torch import nn


class LinearNorm(nn.Module):
    def __init__(self, in_dim, out_dim, bias=True, w_init_gain="linear"):
        super().__init__()
        self.linear_layer = nn.Linear(in_dim, out_dim, bias=bias)
        nn.init.xavier_uniform_(self.linear_layer.weight, gain=nn.init.calculate_gain(w_init_gain))

    def forward(self, x):
        return self.linear_layer(x)


class ConvNorm(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=None, dilation=1, bias=True,
                 w_init_gain="linear"):
        super().__init__()
        if padding is None:
            assert kernel_size % 2 == 1
            padding = int(dilation * (kernel_size - 1) / 2)
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                              dilation=dilation, bias=bias)
        nn.init.xavier_uniform_(self.conv.weight, gain=nn.init.calculate_gain(w_init_gain))

    def forward(self, signal):
        return self.conv(signal)


def __getattr__(name):
    """`from layers import TacotronSTFT` (layers.py:42-80 of the reference): defined in stft.py here; resolved lazily so
    that importing the model does not pull in scipy.signal."""
    if name == "TacotronSTFT":
        from .stft import TacotronSTFT
        return TacotronSTFT
    raise AttributeError(name)
