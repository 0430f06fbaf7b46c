"""Model selector for `main.py -m` (same values and string form as the reference's enums.py)."""
import enum


class ModelEnum(enum.Enum):
    DCGAN = "DCGAN"
    CGAN = "CGAN"

    def __str__(self) -> str:      # argparse shows / parses the bare value
        return str(self.value)
