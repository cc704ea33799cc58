def mobilenet_v2(*args, **kwargs):
    raise RuntimeError("torchvision.models.mobilenet.mobilenet_v2 is not available (shim)")
