"""Tensor fast path for observations.

The reference turns a list of obs dicts into tensors on the host every step:
`RawImagePreprocessor` stacks the images with numpy and `InstructionsPreprocessor` regex-tokenises
EVERY mission string EVERY step, growing a vocabulary in first-seen order
(babyai/utils/format.py:44-82, 100-119).  With the engine the image already is a device tensor and
the mission token ids are produced on the device when an episode starts (k_tokens), so the
preprocessor below never leaves the GPU:

    env = BatchedBabyAIEnv(...); pre = TensorObssPreprocessor(env)
    obs = env.reset(); batch = pre(obs)        # batch.image float32 [N,7,7,3], batch.instr int64 [N,L]

`batch` quacks like `babyai.rl.DictList` (attribute access, len, integer/slice/tensor indexing), which is
what `ACModel.forward` (babyai/model.py:217-273) and `BaseAlgo.collect_experiences` (base.py:131-188)
touch.  The engine's ids are FIXED (the 32 baby-language words, babyai_amd/missions.py VOCAB); a reference vocabulary
(`vocab.json`, ids in first-seen order, format.py:15-41) is honoured through a remap table on the device
(`load_vocab`), and `save_vocab` / `vocab_dict()` write the reference's format.
"""
from .missions import VOCAB, WORD_TO_ID


class TensorDict(dict):
    """Minimal DictList twin (babyai/rl/utils/dictlist.py:1-23)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__

    def __len__(self):
        return len(next(iter(dict.values(self))))

    def __getitem__(self, index):
        if isinstance(index, str):
            return dict.__getitem__(self, index)
        return TensorDict({k: v[index] for k, v in dict.items(self)})


def remap_table(vocab, max_size=100):
    """(lut, vocab'): lut[fixed id] = id in the reference vocabulary `vocab` ({word: id}, format.py:15-41); words it has
    never seen get the next free ids, as `Vocabulary.__getitem__` hands them out (format.py:24-29), in VOCAB order."""
    vocab = {str(k): int(v) for k, v in vocab.items()}
    for w in VOCAB:
        if w not in vocab:
            if len(vocab) >= max_size:
                raise ValueError("Maximum vocabulary capacity reached")
            vocab[w] = len(vocab) + 1
    return [0] + [vocab[w] for w in VOCAB], vocab


class TensorObssPreprocessor(object):
    """`vocab`: None = the fixed ids of missions.VOCAB; or a reference vocabulary -- the `vocab.json` a reference model was
    trained with (`Vocabulary.save`, babyai/utils/format.py:31-35: {word: id} in first-seen order) as a path or dict.
    Token ids then go through a 33-entry remap table ON THE DEVICE, so a reference-trained `ACModel` runs on the tensor
    fast path with the ids it was trained on.  Words the loaded vocabulary has never seen get the next free ids, as
    `Vocabulary.__getitem__` would hand them out (format.py:24-29), in missions.VOCAB order."""

    def __init__(self, env, max_vocab=100, vocab=None):
        self.env = env
        self.tokens = env.enable_instr_tokens()
        self.max_size = max_vocab
        self.obs_space = {"image": 147, "instr": max_vocab}     # same keys as ObssPreprocessor.obs_space
        self.vocab = dict(WORD_TO_ID)
        self.lut = None
        self.width = int(getattr(env, "max_mission_tokens", self.tokens.shape[1]))
        if vocab is not None:
            self.load_vocab(vocab)

    def load_vocab(self, vocab):
        import json
        if not isinstance(vocab, dict):
            with open(vocab) as f:
                vocab = json.load(f)
        lut, vocab = remap_table(vocab, self.max_size)          # fixed id (1..32) -> the loaded vocabulary's id; 0 = padding
        torch = self.env.torch
        self.lut = torch.as_tensor(lut, dtype=torch.int64, device=self.env.device)
        self.vocab = vocab
        return vocab

    def save_vocab(self, path):
        """The vocabulary in the reference's file format (format.py:31-35)."""
        import json
        with open(path, "w") as f:
            json.dump(self.vocab, f)

    def vocab_dict(self):
        return dict(self.vocab)

    @staticmethod
    def words():
        return list(VOCAB)

    def __call__(self, obs=None, device=None, trim=True):
        """trim=True pads to the longest mission of the batch like the reference (one device->host read of the length);
        trim=False uses the level's fixed width (no synchronisation)."""
        torch = self.env.torch
        obs = obs if obs is not None else {"image": self.env.pixels if self.env.pixel else self.env.image}
        image = obs["image"].to(torch.float32)                  # RawImagePreprocessor: float image, no scaling
        tok = self.tokens
        width = max(int((tok != 0).sum(dim=1).max().item()), 1) if trim else self.width
        instr = tok[:, :width].to(torch.int64)
        if self.lut is not None:
            instr = self.lut[instr]
        return TensorDict(image=image, instr=instr)
