def add_prefix(inputs, prefix):
    """{'k': v} -> {'prefix.k': v}  (depth/core/utils/misc.py)."""
    return {f'{prefix}.{name}': value for name, value in inputs.items()}
