import abc


class Trainer(abc.ABC):
    """Common base of the GAN trainers (reference train/trainer.py:4-7): one abstract `train()`."""

    @abc.abstractmethod
    def train(self):
        raise NotImplementedError
