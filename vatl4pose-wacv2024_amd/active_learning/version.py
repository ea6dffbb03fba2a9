__version__ = "0.1.0+mi355x"
