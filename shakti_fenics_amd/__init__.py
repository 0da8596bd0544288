"""MI355X-native SHAKTI subglacial-hydrology solve loop behind the shakti-fenics setup API."""
__version__ = "0.1.0"
