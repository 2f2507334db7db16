"""Vocabulary handling (reference masr/data_utils/featurizer/text_featurizer.py)."""


class TextFeaturizer(object):
    def __init__(self, vocab_filepath):
        self.unk = "<unk>"
        self._vocab_dict, self._vocab_list = self._load_vocabulary_from_file(vocab_filepath)

    def featurize(self, text):
        out = []
        for token in list(text.strip()):
            if token == ' ':
                token = '<space>'
            if token not in self._vocab_dict:
                token = self.unk
            out.append(self._vocab_dict[token])
        return out

    @property
    def vocab_size(self):
        return len(self._vocab_list)

    @property
    def vocab_list(self):
        return self._vocab_list

    @staticmethod
    def _load_vocabulary_from_file(vocab_filepath):
        """One ``token\\tcount`` per line; the index is the line number (text_featurizer.py:52-59)."""
        with open(vocab_filepath, 'r', encoding='utf-8') as f:
            lines = f.readlines()
        vocab_list = [line.split('\t')[0].replace('\n', '') for line in lines]
        return {tok: i for i, tok in enumerate(vocab_list)}, vocab_list
