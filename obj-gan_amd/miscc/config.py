"""Global configuration object `cfg` of the image generator.

Same surface as the reference (reference image_generation/miscc/config.py:9-87): a mutable
attribute-dictionary imported as `from miscc.config import cfg`, with the same keys and default
values, so reference scripts and checkpoints keep working.  easydict is not a dependency here;
`AttrDict` below is the few lines of it that are needed.
"""


class AttrDict(dict):
    """dict whose items are also attributes; nested dicts become AttrDicts."""

    def __init__(self, mapping=None, **kw):
        super().__init__()
        data = dict(mapping or {})
        data.update(kw)
        for k, v in data.items():
            self[k] = v

    def __setitem__(self, key, value):
        if isinstance(value, dict) and not isinstance(value, AttrDict):
            value = AttrDict(value)
        super().__setitem__(key, value)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value


_DEFAULTS = {
    "DATASET_NAME": "coco",
    "DATA_DIR": "",
    "GPU_IDS": "0",
    "CUDA": True,
    "WORKERS": 0,
    "RNN_TYPE": "LSTM",
    "TREE": {"BRANCH_NUM": 3, "BASE_SIZE": 64},
    "TRAIN": {
        "BATCH_SIZE": 64,
        "MAX_EPOCH": 60,
        "SNAPSHOT_INTERVAL": 1,
        "PRINT_INTERVAL": 100,
        "DISPLAY_INTERVAL": 500,
        "DISCRIMINATOR_LR": 2e-4,
        "GENERATOR_LR": 2e-4,
        "FLAG": True,
        "NET_E": "/pretrained/text_encoder100.pth",
        "NET_G": "",
        "BUATTN_NORM": True,
        "SMOOTH": {
            "GAMMA1": 4.0,
            "GAMMA2": 5.0,
            "GAMMA3": 10.0,
            "DAMSM_LAMBDA": 100.0,
            "TXT_LAMBDA": 0.1,
            "SHP_LAMBDA": 1.0,
            "OBJ_LAMBDA": 0.1,
            "UNCOND_LAMBDA": 1.0,
        },
    },
    "TEST": {
        "USE_GT_BOX_SEG": 2,
        "NET_SHP_G": "/pretrained/shape_ckpt/shape_gen.pth",
        "SAVE_OPTIONS": "IMAGE",
        "FID_DIMS": 2048,
        "USE_TF": 1,
        "TEST_IMG_NUM": 1000000,
        "RP_POOL_SIZE": 100,
        "SAMPLE_VAL": False,
    },
    "GAN": {
        "DF_DIM": 96,
        "GF_DIM": 48,
        "Z_DIM": 100,
        "CONDITION_DIM": 100,
        "R_NUM": 1,
        "LOCAL_R_NUM": 3,
        "GLB_R_NUM": 7,
        "LAYER_D_NUM": 4,
    },
    "TEXT": {
        "CAPTIONS_PER_IMAGE": 5,
        "EMBEDDING_DIM": 256,
        "GLOVE_EMBEDDING_DIM": 50,
        "WORDS_NUM": 12,
    },
    "ROI": {
        "BOXES_NUM": 10,
        "BOXES_DIM": 6,
        "FM_SIZE": 16,
        "ROI_MIN_SIZE": 10,
        "BOX_WORDS_NUM": 1,
        "ROI_BASE_SIZE": 5,
        "ROI_SIZE_THRS": 16.0,
    },
}

cfg = AttrDict(_DEFAULTS)
__C = cfg
