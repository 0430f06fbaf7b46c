"""Process-wide logger with the reference's behaviour (logger/main_logger.py): a singleton around logging.getLogger('main'),
stderr handler, optional `<save_path>/<timestamp>.log` file handler when args.log_file == 1, uncaught exceptions logged
through sys.excepthook.  Addition for data-parallel runs: only rank 0 emits unless a message passes gpu_rank explicitly."""
import logging
import os
import sys
from datetime import datetime

_FORMAT = "%(asctime)s %(levelname)s:%(message)s"


class MainLogger:
    _instance = None
    _initialized = False

    def __new__(cls, *args, **kwargs):
        if cls._instance is None:
            cls._instance = super().__new__(cls)
        return cls._instance

    def __init__(self, args=None):
        if MainLogger._initialized:
            return
        self.logger_name = "main"
        self.rank = int(os.environ.get("RANK", "0"))
        self.logger = logging.getLogger(self.logger_name)
        self.logger.setLevel(logging.DEBUG)
        stream = logging.StreamHandler()
        stream.setFormatter(logging.Formatter(_FORMAT))
        self.logger.addHandler(stream)
        if args is not None and getattr(args, "log_file", 0) == 1 and self.rank == 0:
            os.makedirs(args.save_path, exist_ok=True)
            fh = logging.FileHandler(os.path.join(args.save_path, datetime.now().strftime("%Y%m%d_%H%M%S") + ".log"))
            fh.setLevel(logging.DEBUG)
            fh.setFormatter(logging.Formatter(_FORMAT))
            self.logger.addHandler(fh)
        MainLogger._initialized = True

        def _hook(exc_type, exc_value, exc_tb):
            if issubclass(exc_type, KeyboardInterrupt):
                sys.__excepthook__(exc_type, exc_value, exc_tb)
                return
            logging.getLogger("main").error("Unexpected exception.", exc_info=(exc_type, exc_value, exc_tb))
        sys.excepthook = _hook

    def _emit(self, gpu_rank: int) -> bool:
        return self.rank == 0 or gpu_rank == self.rank

    def debug(self, msg, gpu_rank: int = -1):
        if self._emit(gpu_rank):
            self.logger.debug(msg)

    def info(self, msg, gpu_rank: int = -1):
        if self._emit(gpu_rank):
            self.logger.info(msg)

    def warning(self, msg, gpu_rank: int = -1):
        if self._emit(gpu_rank):
            self.logger.warning(msg)

    def error(self, msg, gpu_rank: int = -1):
        if self._emit(gpu_rank):
            self.logger.error(msg)

    def exception(self, msg, gpu_rank: int = -1):
        if self._emit(gpu_rank):
            self.logger.exception(msg)
