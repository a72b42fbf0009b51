"""Empty stand-in for third-party pytorch_lightning (absent here)."""


class Callback:
    pass


class LightningModule:
    pass


class loggers:
    TensorBoardLogger = type('TensorBoardLogger', (), {})
    WandbLogger = type('WandbLogger', (), {})
