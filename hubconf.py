"""torch.hub entry points mirroring the reference's hubconf.py:25-44."""
dependencies = ["torch"]

from demonet_amd.models import ssd_lite_mobilenet_v2, ssdlite320_mobilenet_v3_large, ssd300_vgg16  # noqa: E402,F401
