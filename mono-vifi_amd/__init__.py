"""mono_vifi_amd -- MI355X-native view-synthesis + photometric-loss hot path behind the
Mono-ViFI `layers.py` / `Trainer` API.

Sub-modules are imported lazily; nothing here touches the GPU or loads the HIP library
until an op is called."""
__version__ = "0.1.0"
