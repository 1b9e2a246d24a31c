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
touch.  The vocabulary is FIXED (the 32 baby-language words, babyai_amd/missions.py VOCAB) rather than
first-seen order; `vocab_dict()` returns it in the reference's `Vocabulary.vocab` format
(format.py:15-41) so a model can be trained and saved against it.
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


class TensorObssPreprocessor(object):
    def __init__(self, env, max_vocab=100):
        self.env = env
        self.tokens = env.enable_instr_tokens()
        self.obs_space = {"image": 147, "instr": max_vocab}     # same keys as ObssPreprocessor.obs_space

    @staticmethod
    def vocab_dict():
        return dict(WORD_TO_ID)

    @staticmethod
    def words():
        return list(VOCAB)

    def __call__(self, obs=None, device=None):
        torch = self.env.torch
        obs = obs if obs is not None else {"image": self.env.pixels if self.env.pixel else self.env.image}
        image = obs["image"].to(torch.float32)                  # RawImagePreprocessor: float image, no scaling
        tok = self.tokens
        length = int((tok != 0).sum(dim=1).max().item())        # pad to the longest mission of the batch
        instr = tok[:, :max(length, 1)].to(torch.int64)
        return TensorDict(image=image, instr=instr)
