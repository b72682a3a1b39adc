"""TEST INFRASTRUCTURE ONLY -- stand-in for `torchvision.models.resnet{18,34,50,101,152}`.

The reference's 2-D `resnet18` (pretorched/models/torchvision_models.py:484-492) delegates all
arithmetic to third-party torchvision (`models.resnet18(pretrained=False, num_classes=...)`),
an UNPINNED dependency (requirements.txt:2, setup.py:41) that is neither under /root/reference
nor installed in this image.  This module restates the published canonical ResNet-18
(He et al. 2015; torchvision's layout): conv7x7/2 -> BN -> ReLU -> maxpool3x3/2 ->
4 stages of 2 basic blocks (64/128/256/512, stride-2 stages use a 1x1 conv + BN shortcut) ->
global average pool -> Linear.  Attribute names are the ones `modify_resnets`
(torchvision_models.py:443-464) touches: conv1 bn1 relu maxpool layer1..4 avgpool fc.

resnet50/101/152 use the published bottleneck (1x1 -> 3x3 carrying the stride -> 1x1, expansion 4),
the shape torchvision has always shipped; resnet50 is the per-frame backbone of TRN (trn.py:207).

**Parity unpinned**: the reference holds no test or golden vector for this path, so this
stand-in *is* the oracle for config 1 and is labelled as such wherever it is used.
"""
import torch.nn as nn


class _Basic2d(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False),
                                            nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + idt)


class _Bottleneck2d(nn.Module):
    def __init__(self, cin, width, stride):
        super().__init__()
        cout = 4 * width
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, cout, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False),
                                            nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idt)


class _ResNet2d(nn.Module):
    def __init__(self, depths, num_classes, bottleneck=False):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for i, (width, depth) in enumerate(zip((64, 128, 256, 512), depths)):
            blocks = []
            for j in range(depth):
                stride = 2 if (i > 0 and j == 0) else 1
                if bottleneck:
                    blocks.append(_Bottleneck2d(cin, width, stride))
                    cin = 4 * width
                else:
                    blocks.append(_Basic2d(cin, width, stride))
                    cin = width
            setattr(self, "layer%d" % (i + 1), nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(cin, num_classes)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(self.avgpool(x).flatten(1))


def resnet18(pretrained=False, num_classes=1000):
    assert not pretrained, "no network in this image"
    return _ResNet2d((2, 2, 2, 2), num_classes)


def resnet34(pretrained=False, num_classes=1000):
    assert not pretrained, "no network in this image"
    return _ResNet2d((3, 4, 6, 3), num_classes)


def resnet50(pretrained=False, num_classes=1000):
    assert not pretrained, "no network in this image"
    return _ResNet2d((3, 4, 6, 3), num_classes, bottleneck=True)


def resnet101(pretrained=False, num_classes=1000):
    assert not pretrained, "no network in this image"
    return _ResNet2d((3, 4, 23, 3), num_classes, bottleneck=True)


def resnet152(pretrained=False, num_classes=1000):
    assert not pretrained, "no network in this image"
    return _ResNet2d((3, 8, 36, 3), num_classes, bottleneck=True)


FACTORIES = {"resnet18": resnet18, "resnet34": resnet34, "resnet50": resnet50, "resnet101": resnet101,
             "resnet152": resnet152}
