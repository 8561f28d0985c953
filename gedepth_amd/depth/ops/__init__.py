from .wrappers import resize

__all__ = ['resize']
