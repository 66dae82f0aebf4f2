"""CustomOp dispatch (reference: vllm/model_executor/custom_op.py:7-62): forward_native is the
pure-torch definition, forward_cuda / forward_hip call the native kernels.  On this build the
native path is the only one used on GPU tensors; forward_native is kept as the executable
specification the tests compare against."""
import torch.nn as nn


class CustomOp(nn.Module):

    def __init__(self, *args, **kwargs):
        super().__init__()
        self._forward_method = self.dispatch_forward()

    def forward(self, *args, **kwargs):
        return self._forward_method(*args, **kwargs)

    def forward_native(self, *args, **kwargs):
        raise NotImplementedError

    def forward_cuda(self, *args, **kwargs):
        raise NotImplementedError

    def forward_hip(self, *args, **kwargs):
        return self.forward_cuda(*args, **kwargs)

    def dispatch_forward(self):
        return self.forward_hip
