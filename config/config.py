"""Config — same contract as the reference's config/config.py:6-20: the YAML's top-level sections are
flattened into ONE dict ("hyp") that is handed to the model selection, the loss and the evaluator."""
import yaml


class Config:
    def __init__(self) -> None:
        self.config = {}

    def update_config(self, args):
        for k, v in vars(args).items():
            self.config[k] = v

    def get_config(self, cfg, args=None):
        with open(str(cfg)) as f:
            sections = yaml.safe_load(f)
        for _, section in sections.items():
            self.config.update(section)
        if args:
            self.update_config(args)
        return self.config
