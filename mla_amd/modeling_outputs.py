"""CausalLMOutputWithPast with the MLA extension fields (reference: transformers/modeling_outputs.py:706-713).

Round 6: `logits` and `loss` may be LAZY. The reference computes `lm_head(h).float()` and the shifted cross entropy on every training
step (transformers/models/llama/modeling_llama.py:1255-1269) although the diffusion objective never reads them and the trainer discards
the output object (training/strategies/base_strategy_mla.py:307,334; SURVEY Appendix A #7: "the drop-in may make logits lazy but must be
able to produce them for parity"). With `lazy_lm` given, the 4.6 TFLOP GEMM + the 2.25 GB fp32 logits + the CE pass run on the FIRST
ACCESS of `.logits` / `.loss` (same kernels, same values); an output nobody inspects costs nothing."""
from typing import Callable, Optional, Tuple

import torch


class CausalLMOutputWithPast:
    _FIELDS = ("loss", "logits", "img_pc_contrastive_loss", "tactile_contrastive_loss", "all_logits_for_action", "past_key_values",
               "hidden_states", "attentions")

    def __init__(self, loss: Optional[torch.Tensor] = None, logits: Optional[torch.Tensor] = None,
                 img_pc_contrastive_loss: Optional[torch.Tensor] = None, tactile_contrastive_loss: Optional[torch.Tensor] = None,
                 all_logits_for_action: Optional[torch.Tensor] = None, past_key_values: Optional[Tuple] = None,
                 hidden_states: Optional[Tuple[torch.Tensor, ...]] = None, attentions: Optional[Tuple[torch.Tensor, ...]] = None,
                 lazy_lm: Optional[Callable[[], Tuple[torch.Tensor, Optional[torch.Tensor]]]] = None,
                 readout_hidden: Optional[torch.Tensor] = None, lazy_last: Optional[Callable[[], torch.Tensor]] = None):
        self._loss, self._logits, self._lazy_lm = loss, logits, lazy_lm
        # read-out mode (round 6, opt-in: MLA.readout_rows_only): `readout_hidden` [n, H] = the rows of the final hidden state the
        # caller asked for; the DENSE final hidden state is produced by `lazy_last` on the first access of `hidden_states` (or of
        # logits / loss, whose closure calls it) -- the trainer never does (base_strategy_mla.py:307,334)
        self.readout_hidden, self._lazy_last = readout_hidden, lazy_last
        self.img_pc_contrastive_loss = img_pc_contrastive_loss
        self.tactile_contrastive_loss = tactile_contrastive_loss
        self.all_logits_for_action = all_logits_for_action
        self.past_key_values = past_key_values
        self._hidden_states = hidden_states
        self.attentions = attentions

    def _materialise(self):
        """Runs lm_head + cross entropy once; `loss` = CE + the contrastive terms, added in the reference's order (:1272-1303)."""
        if self._lazy_lm is not None:
            fn, self._lazy_lm = self._lazy_lm, None
            self._logits, loss = fn()
            if loss is not None:
                for extra in (self.img_pc_contrastive_loss, self.tactile_contrastive_loss):
                    if extra is not None:
                        loss = loss + extra
            self._loss = loss

    @property
    def hidden_states(self):
        if self._lazy_last is not None and self._hidden_states is not None:
            fn, self._lazy_last = self._lazy_last, None
            self._hidden_states = tuple(self._hidden_states) + (fn(),)
        return self._hidden_states

    @hidden_states.setter
    def hidden_states(self, v):
        self._hidden_states, self._lazy_last = v, None

    @property
    def last_hidden_pending(self) -> bool:
        """True while the dense final hidden state of a read-out forward has not been asked for."""
        return self._lazy_last is not None

    @property
    def lm_head_pending(self) -> bool:
        """True while logits / loss have not been asked for (bench.py reports `lm_head: lazy` and does not count their flops)."""
        return self._lazy_lm is not None

    @property
    def logits(self):
        self._materialise()
        return self._logits

    @logits.setter
    def logits(self, v):
        self._logits = v

    @property
    def loss(self):
        self._materialise()
        return self._loss

    @loss.setter
    def loss(self, v):
        self._loss = v

    def __getitem__(self, k):
        if isinstance(k, str):
            return getattr(self, k)
        return tuple(v for v in (self.loss, self.logits, self.past_key_values, self.hidden_states, self.attentions)
                     if v is not None)[k]

    def __repr__(self):
        def show(f):
            if (f in ("loss", "logits") and self.lm_head_pending) or (f == "hidden_states" and self.last_hidden_pending):
                return "<lazy>"          # (printing an output must not run lm_head or the dense last layer)
            return type(getattr(self, f)).__name__
        return "CausalLMOutputWithPast(" + ", ".join(f"{f}={show(f)}" for f in self._FIELDS) + ")"
