def paste_masks_in_image(*args, **kwargs):
    raise RuntimeError("masks are not part of the SSD path (shim)")
