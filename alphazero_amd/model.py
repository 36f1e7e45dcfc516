"""Policy/value network of the reference (src/model.py:9-44, `simplerNN`), restated.

A plain convolutional stack (not residual): a shared trunk of 3x3 conv + batch
norm + ReLU blocks, a policy head ending in a 3x3 conv to one channel with
log-softmax over the board, and a value head ending in a 1x1 conv, a board-wide
average and tanh.  Sub-module names (`front`, `policy_conv`, `policy`,
`value_conv`, `value`) and their Sequential indices match the reference so its
`state_dict` checkpoints load unchanged; defaults are the reference's
config (general_config.py:108-129).  Dense conv work is left to MIOpen/rocBLAS
(MFMA lives there); the tree kernels never touch it.
"""
import torch
from torch import nn


def _blocks(c_in, c_out, count):
    layers = []
    for i in range(count):
        layers += [nn.Conv2d(c_in if i == 0 else c_out, c_out, 3, padding=1, bias=False),
                   nn.BatchNorm2d(c_out), nn.ReLU()]
    return nn.Sequential(*layers)


class PolicyValueNet(nn.Module):
    def __init__(self, shape=13, channels=2, conv_channels=16, front_layers=4, policy_layers=2, value_layers=2):
        super().__init__()
        self.shape = shape
        self.front = _blocks(channels, conv_channels, front_layers)
        self.policy_conv = _blocks(conv_channels, conv_channels, policy_layers)
        self.policy = nn.Sequential(nn.Conv2d(conv_channels, 1, 3, padding=1))
        self.value_conv = _blocks(conv_channels, conv_channels, value_layers)
        self.value = nn.Sequential(nn.Conv2d(conv_channels, 1, 1), nn.AvgPool2d(shape), nn.Tanh())

    @classmethod
    def from_config(cls, conf, shape):
        """conf: an object with the reference's config names (CHANNELS, CONV_CHANNELS, ...)."""
        return cls(shape, conf.CHANNELS, conf.CONV_CHANNELS, conf.FRONT_LAYER_CNT,
                   conf.POLICY_LAYER_CNT, conf.VALUE_LAYER_CNT)

    def forward(self, boards):
        trunk = self.front(boards)
        logits = self.policy(self.policy_conv(trunk)).flatten(1)
        return self.value(self.value_conv(trunk)), torch.log_softmax(logits, dim=1)
