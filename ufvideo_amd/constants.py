"""Sentinel ids and defaults of the hot path (values as in the reference's ufvideo/constants.py:7-57;
they are observable behaviour: the splice keys on them)."""
IGNORE_INDEX = -100

IMAGE_TOKEN_INDEX = -200
DEFAULT_IMAGE_TOKEN = "<image>"
VIDEO_TOKEN_INDEX = -201
DEFAULT_VIDEO_TOKEN = "<video>"
AUDIO_TOKEN_INDEX = -202
DEFAULT_AUDIO_TOKEN = "<audio>"

NUM_FRAMES = 32
MAX_FRAMES = 32
NUM_FRAMES_PER_SECOND = 1

TEMPORAL_TOKEN_FORMAT = "<TEMP-{:03d}>"

MODAL_INDEX_MAP = {
    DEFAULT_IMAGE_TOKEN: IMAGE_TOKEN_INDEX,
    DEFAULT_VIDEO_TOKEN: VIDEO_TOKEN_INDEX,
    DEFAULT_AUDIO_TOKEN: AUDIO_TOKEN_INDEX,
}

QUESTION_LIST = [
    "Can you segment the {class_name} in this image?",
    "Please segment the {class_name} in this image.",
    "What is {class_name} in this image? Please respond with segmentation mask.",
    "What is {class_name} in this image? Please output segmentation mask.",
]

ANSWER_LIST = [
    "It is [SEG].",
    "Sure, [SEG].",
    "Sure, it is [SEG].",
    "Sure, the segmentation result is [SEG].",
    "[SEG].",
]
