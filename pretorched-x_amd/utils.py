"""`Identity`, the module the reference's README tells users to put in place of `last_linear` for feature extraction
(`model.last_linear = pretrained.utils.Identity()`, /root/reference/README.md:543-546; defined at
/root/reference/pretorched/models/utils.py:81-87 and re-exported as `pretorched.models.Identity`, models/__init__.py:79).
The engine reads `last_linear` at call time: anything that is not a plain fp32 nn.Linear on the model's device is simply
CALLED on the pooled features (engine.py `Plan.run_head`), so this class needs no special casing."""
import torch


class Identity(torch.nn.Module):

    def __init__(self):
        super(Identity, self).__init__()

    def forward(self, x):
        return x
