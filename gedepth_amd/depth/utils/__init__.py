from .position_encoding import SinePositionalEncoding

__all__ = ['SinePositionalEncoding']
