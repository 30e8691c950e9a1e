"""Config — same contract as the reference's config/config.py:6-20: the YAML's top-level sections are
flattened into ONE dict ("hyp") that is handed to the model selection, the loss and the evaluator; attributes of an
optional argparse namespace override the file."""
import yaml


class Config:
    def __init__(self) -> None:
        self.config = {}

    def update_config(self, args):
        self.config.update(vars(args))

    def get_config(self, cfg, args=None):
        with open(str(cfg)) as stream:
            for section in (yaml.safe_load(stream) or {}).values():
                if isinstance(section, dict):
                    self.config.update(section)
        if args is not None:
            self.update_config(args)
        return self.config
