"""``$ROOT`` expansion of the entry scripts' path flags (reference ``utils/update_paths.py:6-24``): ``$ROOT`` is the directory
that holds the package (the repository root), so the reference's default locations (``$ROOT/data/...``) keep their meaning."""
import os

__all__ = ["update_paths"]


def update_paths(config):
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
    for key, value in vars(config).items():
        if isinstance(value, str) and "$ROOT" in value:
            setattr(config, key, value.replace("$ROOT", root).replace("/", os.sep))
