"""``KGC`` data module and batch collator behind the reference's ``data.data_module`` names (MarT/data/data_module.py).

Reference behaviour kept (line refs to MarT/data/data_module.py):
  * ``KGC(args, model)`` (:184-233): tokenizer + 11 292 ``[ENTITY_i]`` then 192 ``[RELATION_j]`` added tokens, the id
    ranges ``entity_id_st/ed``, ``relation_id_st/ed`` and the analogy id lists that ``TransformerLitModel`` reads through
    ``get_config()`` (:245-251, same substring filter); ``setup()`` builds the three splits with ``get_dataset``;
    ``{train,val,test}_dataloader()`` with shuffle only for train and ``eval_batch_size`` for the others (:267-274).
  * ``DataCollatorForSeq2Seq.__call__`` (:91-182): pops the per-example fields, right-pads the three token arrays with
    ``tokenizer.pad`` (longest of the batch, multiple of 8 under ``precision == 16``), and picks the two image slots per
    example -- both entities present: (head, tail); otherwise entity = head if head is not None else tail:
    (entity or zeros, zeros) (:126-142).

MI355X-side design: the image table ``[N,3,224,224]`` stays resident in HBM
(``model.set_image_table``); the collator only emits ``image_index [B,2] int32`` (-1 = zero image) through a dict lookup
instead of ``list.index`` + stacking 1.2 MB/example on the host, and the patch gather happens in
``mart_patchify_gather``.  Passing ``visual_features`` (a CPU tensor) instead reproduces the reference's
``pixel_values [B,2,3,H,W]`` output literally (used by the parity tests).  Feature dicts are not mutated (the
reference's ``pop`` empties the dataset's own dicts, which only works once per process).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Union

import torch
from torch.utils.data import DataLoader

from ..batching import DeviceImageTable
from .processor import KGProcessor, get_dataset
from .tokenization import BertWordPieceTokenizer

BATCH_SIZE = 8
NUM_WORKERS = 8
_TOKEN_KEYS = ("input_ids", "attention_mask", "token_type_ids")
_POPPED = ("label", "rel_label", "head_ent", "tail_ent", "pre_type", "rel_idx", "sep_idx", "q_head_idx", "a_head_idx")


class Config(dict):
    def __getattr__(self, name):
        return self.get(name)

    def __setattr__(self, name, val):
        self[name] = val


class BaseDataModule:
    """MarT/data/base_data_module.py:23-70 without the LightningDataModule parent (PL is not a dependency here)."""

    def __init__(self, args=None) -> None:
        self.args = Config(vars(args)) if args is not None else {}
        self.batch_size = self.args.get("batch_size", BATCH_SIZE)
        self.num_workers = self.args.get("num_workers", NUM_WORKERS)

    @staticmethod
    def add_to_argparse(parser):
        parser.add_argument("--batch_size", type=int, default=BATCH_SIZE, help="Number of examples to operate on per forward step.")
        parser.add_argument("--num_workers", type=int, default=0, help="Number of additional processes to load data.")
        parser.add_argument("--dataset", type=str, default="./dataset/NELL", help="Dataset directory.")
        return parser

    def prepare_data(self):
        pass

    def setup(self, stage=None):
        self.data_train = self.data_val = self.data_test = None


@dataclass
class DataCollatorForSeq2Seq:
    tokenizer: Any
    model: Optional[Any] = None
    padding: Union[bool, str] = True
    max_length: Optional[int] = None
    pad_to_multiple_of: Optional[int] = None
    label_pad_token_id: int = -100
    return_tensors: str = "pt"
    num_labels: int = 0
    image_table: Optional[DeviceImageTable] = None        # entity -> row of the HBM-resident image table
    visual_features: Optional[torch.Tensor] = None         # CPU [N,3,H,W]: emit pixel_values like the reference

    def __call__(self, features: Sequence[Dict[str, Any]], return_tensors=None) -> Dict[str, Any]:
        return_tensors = self.return_tensors if return_tensors is None else return_tensors
        first = features[0]
        col = {k: [f[k] for f in features] for k in _POPPED if k in first}
        extra = {k: [f[k] for f in features] for k in first if k not in _TOKEN_KEYS and k not in _POPPED}
        batch = self.tokenizer.pad([{k: f[k] for k in _TOKEN_KEYS if k in f} for f in features], padding=self.padding,
                                   max_length=self.max_length, pad_to_multiple_of=self.pad_to_multiple_of,
                                   return_tensors=return_tensors)
        batch = dict(batch)
        head_ent, tail_ent = col.get("head_ent"), col.get("tail_ent")
        if head_ent is not None and self.image_table is not None:
            slots = self.image_table.slots(head_ent, tail_ent)                       # [B,2] int32, -1 = zero image
            if self.visual_features is not None:
                tab = self.visual_features
                zero = torch.zeros(tab.shape[1:], dtype=tab.dtype)
                batch["pixel_values"] = torch.stack([torch.stack([tab[i] if i >= 0 else zero for i in row.tolist()])
                                                     for row in slots])
            else:
                batch["image_index"] = slots
        batch["label"] = torch.tensor(col["label"])
        # The reference fails loudly on a prompt without exactly one [MASK] (transformer.py:74-75 assert / shape mismatch at :95)
        # and torch raises on an out-of-range label; the device kernels index without checks, so both are validated here, on
        # the host, where the batch is built (a prompt truncated by max_seq_length loses its trailing [MASK]).
        mask_id = getattr(self.tokenizer, "mask_token_id", None)
        if mask_id is not None and "input_ids" in batch and torch.is_tensor(batch["input_ids"]):
            n_mask = (batch["input_ids"] == mask_id).sum(1)
            if not bool((n_mask == 1).all()):
                raise ValueError(f"every prompt needs exactly one [MASK]; rows {torch.nonzero(n_mask != 1).flatten().tolist()} have {n_mask[n_mask != 1].tolist()}")
        if bool((batch["label"] < 0).any()) or (self.num_labels and bool((batch["label"] >= self.num_labels).any())):
            raise ValueError("label out of range (ignore_index / negative labels are not supported by the fused loss)")
        for k in ("pre_type", "rel_idx", "sep_idx", "rel_label", "q_head_idx", "a_head_idx"):
            if col.get(k):
                batch[k] = torch.tensor(col[k])
        batch.update(extra)
        return batch


class KGC(BaseDataModule):
    def __init__(self, args, model=None, tokenizer=None, visual_features: Optional[torch.Tensor] = None) -> None:
        """``tokenizer`` / ``visual_features`` may be injected (tests, synthetic runs); by default they are read from
        ``args.model_name_or_path/vocab.txt`` and ``args.data_dir/entity_image_features.CLIP-VIT-16-32.pth`` (:207)."""
        super().__init__(args)
        a = self.args
        self.tokenizer = tokenizer if tokenizer is not None else \
            BertWordPieceTokenizer.from_pretrained(a.model_name_or_path, use_fast=False)
        self._fresh_len = len(self.tokenizer)                   # processor.py:256 builds features with a fresh tokenizer
        self.processor = KGProcessor(self.tokenizer, a)
        self.label_list = self.processor.get_labels(a.data_dir)
        entity_list = self.processor.get_entities(a.data_dir)
        self.tokenizer.add_special_tokens({"additional_special_tokens": entity_list})
        with open(self.processor.entity_path, "r") as f:
            self.entities = [line.strip().split("\t")[0] for line in f.readlines()]
        if visual_features is None:
            if a.model_class in ("VisualBertKGC", "VilBertKGC", "ViltKGC"):
                raise NotImplementedError(f"{a.model_class} is outside the MKGformer/FLAVA hot path (SURVEY 8: out of scope)")
            path = os.path.join(a.data_dir, "entity_image_features.CLIP-VIT-16-32.pth")
            if not os.path.exists(path):
                raise FileNotFoundError(f"{path} not found: pass visual_features=[N,3,224,224] or provide the file")
            visual_features = torch.load(path)
        self.visual_features = visual_features
        self.image_table = DeviceImageTable(self.entities)
        self.emit_pixel_values = False                         # True: reference-literal pixel_values on the host
        relations_tokens = self.processor.get_relations(a.data_dir)
        self.num_relations = len(relations_tokens)
        self.tokenizer.add_special_tokens({"additional_special_tokens": relations_tokens})
        vocab = self.tokenizer.get_added_vocab()
        self.relation_id_st = vocab[relations_tokens[0]]
        self.relation_id_ed = vocab[relations_tokens[-1]] + 1
        self.entity_id_st = vocab[entity_list[0]]
        self.entity_id_ed = vocab[entity_list[-1]] + 1
        self.analogy_entity_ids = [vocab[e] for e in self.processor.get_analogy_entities(a.data_dir)]
        self.analogy_relation_ids = [vocab[r] for r in self.processor.get_analogy_relations(a.data_dir)]
        self.sampler = self._make_sampler(model)

    def _make_sampler(self, model=None) -> DataCollatorForSeq2Seq:
        a = self.args
        return DataCollatorForSeq2Seq(self.tokenizer, model=model, label_pad_token_id=self.tokenizer.pad_token_id,
                                      pad_to_multiple_of=8 if a.precision == 16 else None, padding="longest",
                                      max_length=a.max_seq_length, num_labels=len(self.entities),
                                      image_table=self.image_table,
                                      visual_features=self.visual_features if self.emit_pixel_values else None)

    def use_host_pixels(self, flag: bool = True) -> None:
        self.emit_pixel_values = flag
        self.sampler = self._make_sampler(self.sampler.model)

    def attach(self, model) -> None:
        """Upload the image table once; batches then carry ``image_index`` only."""
        model.set_image_table(self.visual_features)

    def _fresh_tokenizer(self):
        """A view of the tokenizer as it was before the entity / relation tokens were added (len 30522)."""
        t = self.tokenizer
        fresh = BertWordPieceTokenizer(t.vocab, do_lower_case=t.do_lower_case, name_or_path=t.name_or_path)
        return fresh

    def setup(self, stage=None):
        fresh = self._fresh_tokenizer()
        self.data_train = get_dataset(self.args, self.processor, "train", fresh)
        self.data_val = get_dataset(self.args, self.processor, "dev", fresh)
        self.data_test = get_dataset(self.args, self.processor, "test", fresh)

    def get_config(self) -> Dict[str, Any]:
        return {k: v for k, v in self.__dict__.items() if "st" in k or "ed" in k or "analogy" in k}    # :245-251

    @staticmethod
    def add_to_argparse(parser):
        BaseDataModule.add_to_argparse(parser)
        parser.add_argument("--model_name_or_path", type=str, default="roberta-base")
        parser.add_argument("--data_dir", type=str, default="roberta-base")
        parser.add_argument("--max_seq_length", type=int, default=256)
        parser.add_argument("--warm_up_radio", type=float, default=0.1)
        parser.add_argument("--eval_batch_size", type=int, default=8)
        parser.add_argument("--overwrite_cache", action="store_true", default=False)
        return parser

    def get_tokenizer(self):
        return self.tokenizer

    def _loader(self, data, batch_size, shuffle):
        """One process per GPU: under an initialised process group every rank reads its own shard (what PL injects into the
        reference's loaders under DDP: DistributedSampler, padded to equal length); Trainer.fit calls set_epoch per epoch."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from torch.utils.data.distributed import DistributedSampler
            ds = DistributedSampler(data, shuffle=shuffle)
            return DataLoader(data, num_workers=self.num_workers, pin_memory=False, collate_fn=self.sampler,
                              batch_size=batch_size, sampler=ds)
        return DataLoader(data, num_workers=self.num_workers, pin_memory=False, collate_fn=self.sampler,
                          batch_size=batch_size, shuffle=shuffle)

    def train_dataloader(self):
        return self._loader(self.data_train, self.args.batch_size, True)

    def val_dataloader(self):
        return self._loader(self.data_val, self.args.eval_batch_size, False)

    def test_dataloader(self):
        return self._loader(self.data_test, self.args.eval_batch_size, False)
