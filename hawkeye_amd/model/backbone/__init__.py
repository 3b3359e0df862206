from .vgg import vgg16, VGG  # noqa: F401
from .resnet import resnet50, resnet101, ResNet, Bottleneck  # noqa: F401
