def load_state_dict_from_url(*args, **kwargs):
    raise RuntimeError("no network in this container: pretrained weights are unavailable")
