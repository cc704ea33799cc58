"""VGG16 'D' feature stack as published by torchvision (cfg D: 13 conv3x3+ReLU, 5 maxpool 2x2)."""
import torch.nn as nn

_CFG_D = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']


class _VGG(nn.Module):
    def __init__(self):
        super().__init__()
        layers = []
        c = 3
        for v in _CFG_D:
            if v == 'M':
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(c, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                c = v
        self.features = nn.Sequential(*layers)


def vgg16(pretrained=False, progress=True, **kwargs):
    assert not pretrained
    return _VGG()
