"""Special token ids and their surface forms (reference transformer/Constants.py:1-9)."""
PAD, UNK, BOS, EOS = 0, 1, 2, 3

PAD_FLAG, UNK_FLAG, BOS_FLAG, EOS_FLAG = '<pad>', '<unk>', '<sos>', '<eos>'
