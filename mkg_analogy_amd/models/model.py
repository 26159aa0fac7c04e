"""Operator classes selected by name from the CLI, as in MarT/models/model.py:7-11."""
from .modeling_unimo import UnimoForMaskedLM


class MKGformerKGC(UnimoForMaskedLM):
    @staticmethod
    def add_to_argparse(parser):
        parser.add_argument("--pretrain", type=int, default=0, help="")
        return parser


class FlavaKGC(__import__("mkg_analogy_amd.models.modeling_flava", fromlist=["FlavaForMaskedLM"]).FlavaForMaskedLM):
    """MarT/models/model.py:31."""

    @staticmethod
    def add_to_argparse(parser):
        parser.add_argument("--pretrain", type=int, default=0, help="")
        return parser
