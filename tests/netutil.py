"""Re-export of the product-side ctypes helpers for the tests."""
from darknet_amd.netapi import *  # noqa: F401,F403
from darknet_amd.netapi import DkNet, cfg_path, synth_weights_for  # noqa: F401
