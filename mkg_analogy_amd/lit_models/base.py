"""Trainer-surface base class (reference: MarT/lit_models/base.py:20-95) without a pytorch_lightning dependency.

``LightningModuleLite`` provides the handful of LightningModule facilities the MarT lit model touches
(``save_hyperparameters``, ``log``, ``trainer``); ``mkg_analogy_amd.trainer.Trainer`` drives the same hooks.
"""
import argparse

import torch

OPTIMIZER = "AdamW"
LR = 5e-5


class Config(dict):
    def __getattr__(self, name):
        return self.get(name)

    def __setattr__(self, name, val):
        self[name] = val


class LightningModuleLite(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.trainer = None
        self.logged = {}

    def save_hyperparameters(self, *a, **k):
        pass

    def log(self, name, value, **kwargs):
        self.logged[name] = float(value)


class BaseLitModel(LightningModuleLite):
    def __init__(self, model, args: argparse.Namespace = None):
        super().__init__()
        self.model = model
        self.args = Config(vars(args)) if args is not None else {}
        self.optimizer_name = self.args.get("optimizer", OPTIMIZER)
        self.lr = self.args.get("lr", LR)

    @staticmethod
    def add_to_argparse(parser):
        parser.add_argument("--optimizer", type=str, default=OPTIMIZER, help="optimizer class (AdamW is the fused HIP implementation)")
        parser.add_argument("--lr", type=float, default=LR)
        parser.add_argument("--weight_decay", type=float, default=0.01)
        return parser

    def forward(self, x):
        return self.model(x)

    @property
    def num_training_steps(self) -> int:
        """(len(train_loader) // (accumulate * devices)) * max_epochs, capped by max_steps (base.py:74-95)."""
        t = self.trainer
        if t is None:
            raise RuntimeError("num_training_steps needs an attached trainer")
        dataset_size = t.num_train_batches
        eff = max(1, t.accumulate_grad_batches) * max(1, t.world_size)
        est = (dataset_size // eff) * t.max_epochs
        if t.max_steps and t.max_steps < est:
            return t.max_steps
        return est
