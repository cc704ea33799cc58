from . import utils, _utils, mobilenet, vgg, detection  # noqa: F401
