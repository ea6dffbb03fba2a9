__version__ = "0.1.0+mi355x"
short_version = "0.1.0"
