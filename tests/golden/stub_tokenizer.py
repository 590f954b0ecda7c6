"""A deterministic stand-in for the HF tokenizer interface that ``tokenizer_image_token`` uses (``tokenizer(text).input_ids``,
``bos_token_id``): the prompt-construction goldens are generated with it through the REFERENCE's functions
(tests/golden/make_golden.py gen_prompt) and replayed through ``interactvlm_amd.demo`` in tests/test_demo.py.  No tokenizer model ships
with the reference (LLaMA's sentencepiece file is a gated download); what the goldens pin is the prompt TEXT and the splice logic
around the image placeholder, which do not depend on the vocabulary."""
import zlib


class StubTokenizer:
    bos_token_id = 1

    def __init__(self, bos: bool = True):
        self.bos = bos

    def __call__(self, text):
        ids = [3 + zlib.crc32(w.encode()) % 30000 for w in text.split()]

        class Enc:
            input_ids = ([self.bos_token_id] if self.bos else []) + ids

        return Enc
