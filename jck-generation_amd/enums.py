"""Model selector for `main.py -m`: members DCGAN and CGAN whose value and string form are their own name, so argparse
can both parse `-m DCGAN` (`type=ModelEnum`) and print the choices bare (the reference's enums.py has the same surface)."""
import enum


class _NamedByValue(enum.Enum):
    def __str__(self) -> str:
        return str(self.value)


MODEL_NAMES = ("DCGAN", "CGAN")
ModelEnum = _NamedByValue("ModelEnum", [(name, name) for name in MODEL_NAMES], module=__name__)
