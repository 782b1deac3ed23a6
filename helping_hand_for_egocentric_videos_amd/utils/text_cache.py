"""Tokenised-text cache (second half): the reference tokenises every caption string on the host in every
step (/root/reference/run/train.py:66,72,74-75 -- `tokenizer(data['text'])`, 5 rephrases per clip plus the EgoClip narration) and
uploads the ids.  Captions repeat across epochs, so their token rows are kept in a device-resident table: a step tokenises only the
strings it has not seen, uploads those rows once, and gathers the batch's [len, context_length] id tensor on the device.

The tokenizer itself is out of scope: any callable `list[str] -> LongTensor [len, context_length]` works (the
reference's SimpleTokenizer, model/tokenizer.py).  Results are bit-identical to calling the tokenizer on the batch."""
import torch


class TokenizedTextCache:
    def __init__(self, tokenizer, context_length=77, device="cuda", capacity=1 << 16):
        self.tokenizer, self.context_length, self.device = tokenizer, context_length, torch.device(device)
        self.index = {}                                               # caption -> row of the table
        self.table = torch.zeros((capacity, context_length), dtype=torch.int64, device=self.device)
        self.hits = self.misses = 0

    def __len__(self):
        return len(self.index)

    def __call__(self, texts):
        """list[str] -> int64 [len(texts), context_length] on the cache's device."""
        new = [t for t in dict.fromkeys(texts) if t not in self.index]
        if new:
            rows = self.tokenizer(new)
            if rows.shape != (len(new), self.context_length):
                raise ValueError("TokenizedTextCache: tokenizer returned %s for %d strings" % (tuple(rows.shape), len(new)))
            start = len(self.index)
            if start + len(new) > self.table.shape[0]:                # grow geometrically (a reallocation, not per step)
                grown = torch.zeros((max(2 * self.table.shape[0], start + len(new)), self.context_length), dtype=torch.int64, device=self.device)
                grown[:start] = self.table[:start]
                self.table = grown
            self.table[start:start + len(new)] = rows.to(device=self.device, dtype=torch.int64, non_blocking=True)
            for i, t in enumerate(new):
                self.index[t] = start + i
        self.misses += len(new)
        self.hits += len(texts) - len(new)
        idx = torch.tensor([self.index[t] for t in texts], dtype=torch.int64)
        if self.device.type == "cuda":
            idx = idx.pin_memory().to(self.device, non_blocking=True)
        return self.table.index_select(0, idx)
