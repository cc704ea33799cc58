"""demonet_amd: MI355X-native (gfx950) SSD inference path behind demonet's model-factory API."""
from . import models  # noqa: F401
from .models import *  # noqa: F401,F403
