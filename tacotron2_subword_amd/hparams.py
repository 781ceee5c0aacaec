"""Hyper-parameter surface of the reference (hparams.py:1-116): same field names, same
defaults, same ad-hoc override-string grammar, so train.py / inference.py callers keep working."""


class AttrDict(dict):
    """dict whose items are also attributes (hparams.py:1-4)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self


_DEFAULTS = dict(
    # experiment (hparams.py:14-24)
    epochs=1500, iters_per_checkpoint=1000, seed=1234, dynamic_loss_scaling=True, fp16_run=False,
    distributed_run=False, dist_backend="nccl", dist_url="tcp://localhost:14897", cudnn_enabled=True,
    cudnn_benchmark=False, ignore_layers=["embedding.weight"],
    # data (hparams.py:30-45)
    load_mel_from_disk=False, load_phone_from_disk=True, datafiles="data/vi_dataset",
    training_files="data/vi_dataset/script/train.txt", validation_files="data/vi_dataset/script/val.txt",
    training_preprocess="data/vi_dataset/preprocess/train.txt", validation_preprocess="data/vi_dataset/preprocess/val.txt",
    bert_embeddings_train_path="bert_embeddings/train", bert_embeddings_val_path="bert_embeddings/val",
    bert_embeddings_cls_train_path="bert_embeddings_cls/train", bert_embeddings_cls_val_path="bert_embeddings_cls/val",
    text_cleaners=["basic_cleaners"],
    # audio (hparams.py:50-57)
    max_wav_value=32768.0, sampling_rate=22050, filter_length=1024, hop_length=256, win_length=1024,
    n_mel_channels=80, mel_fmin=0.0, mel_fmax=8000.0,
    # model (hparams.py:62-95)
    n_symbols=313, sub_n_symbols=5500, symbols_embedding_dim=512, alignloss="", attention="StepwiseMonotonicAttention",
    encoder_kernel_size=5, encoder_n_convolutions=3, encoder_embedding_dim=512, BERT_embedding_dim=768,
    n_frames_per_step=1, decoder_rnn_dim=1024, prenet_dim=256, max_decoder_steps=1000, gate_threshold=0.001,
    p_attention_dropout=0.1, p_decoder_dropout=0.1, attention_rnn_dim=1024, attention_dim=128,
    attention_location_n_filters=32, attention_location_kernel_size=31,
    postnet_embedding_dim=512, postnet_kernel_size=5, postnet_n_convolutions=5,
    # optimisation (hparams.py:100-105)
    use_saved_learning_rate=True, learning_rate=1e-3, weight_decay=1e-6, grad_clip_thresh=1.0, batch_size=8,
    mask_padding=True,
)


def create_hparams(hparams_string=None, verbose=False):
    """Defaults + the reference's override grammar (hparams.py:108-114): one leading and two
    trailing characters are stripped, pairs are '-'-separated 'key:value', values stay strings."""
    hp = AttrDict({k: (list(v) if isinstance(v, list) else v) for k, v in _DEFAULTS.items()})
    if hparams_string:
        for pair in hparams_string[1:-2].split("-"):
            k, v = pair.split(":")
            if k in hp:
                hp[k] = v
                print("Set hparam: " + k + " to " + v)
    return hp
