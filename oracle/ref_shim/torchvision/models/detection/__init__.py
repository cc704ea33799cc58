from . import image_list, roi_heads  # noqa: F401
