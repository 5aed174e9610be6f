"""Terminal colouring helper with the reference's call shape: ``TextColors.colorize(text, "cyan")``."""
from enum import Enum


class _Ansi(Enum):
    black, red, green, yellow, blue, magenta, cyan, white = range(30, 38)

    @property
    def code(self) -> str:
        return f"\x1b[{self.value}m"


_RESET = "\x1b[0m"


class TextColors:
    #: name -> escape sequence (kept as a mapping because callers index it)
    COLORS = {**{c.name: c.code for c in _Ansi}, "reset": _RESET}

    @classmethod
    def colorize(cls, text, color: str) -> str:
        """Wrap `text` in the colour's escape codes; unknown colours leave it unstyled."""
        prefix = cls.COLORS.get(str(color).lower(), _RESET)
        return "".join((prefix, str(text), _RESET))


def load_config(path: str) -> dict:
    """Engine configuration file -> kwargs dict.  ``.json`` (the reference's format) or ``.yaml`` / ``.yml``."""
    import json
    with open(path) as f:
        if path.endswith((".yaml", ".yml")):
            import yaml
            return dict(yaml.safe_load(f))
        return json.load(f)
