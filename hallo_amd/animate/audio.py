"""AudioProcessor: waveform -> per-frame wav2vec2 embedding, the step immediately before the denoising path
(SURVEY.md section 8f row 2).

Reference: hallo/datasets/audio_processor.py:30-129.  Kept: the constructor's meaning of (sample_rate, fps, wav2vec
model, only_last_features), `preprocess(wav_file, clip_length)` -> (audio_emb fp32 CPU [seq_len, 12, 768], audio_length),
the zero-mean / unit-variance normalisation of Wav2Vec2FeatureExtractor, seq_len = ceil(len / sample_rate * fps), zero
padding of the waveform up to a multiple of clip_length frames, hidden_states[1:] stacked as "s b d".
Out of scope (I/O and third-party preprocessing): the vocal separator (audio_separator) and librosa's
resampling loader; `preprocess` reads 16-bit / float PCM WAV files that are already at `sample_rate` (the reference's own
`resample_audio` step produces exactly such a file with ffmpeg) and `preprocess_array` takes the loaded array.
"""
import math
import wave

import numpy as np
import torch

from ..models.wav2vec import Wav2VecModel


def load_wav(path, sample_rate):
    """Mono float32 array in [-1, 1) from a PCM WAV file recorded at `sample_rate` (channels are averaged, as
    librosa.load(mono=True) does)."""
    with wave.open(path, "rb") as f:
        sr, nch, width, n = f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()
        raw = f.readframes(n)
    if sr != sample_rate:
        raise ValueError(f"{path} is sampled at {sr} Hz; resample it to {sample_rate} Hz first "
                         "(the reference does so with ffmpeg in resample_audio, hallo/datasets/audio_processor.py:99)")
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"unsupported sample width {width} bytes in {path}")
    if nch > 1:
        x = x.reshape(-1, nch).mean(axis=1)
    return x


class AudioProcessor:
    def __init__(self, sample_rate, fps, wav2vec_model, only_last_features=False, device="cuda:0", dtype=torch.float16):
        """`wav2vec_model`: a hallo_amd Wav2VecModel (already on the GPU) or the checkpoint directory the reference passes
        as `wav2vec_model_path` (config.json + weights; loaded locally, never downloaded)."""
        self.sample_rate, self.fps, self.only_last_features = sample_rate, fps, only_last_features
        if isinstance(wav2vec_model, Wav2VecModel):
            self.audio_encoder = wav2vec_model
        else:
            self.audio_encoder = Wav2VecModel.from_pretrained(wav2vec_model, local_files_only=True).to(device, dtype)
        self.audio_encoder.feature_extractor._freeze_parameters()
        self.device = self.audio_encoder.device

    @staticmethod
    def normalize(speech_array):
        """Wav2Vec2FeatureExtractor(do_normalize=True): (x - mean) / sqrt(var + 1e-7), per utterance, fp32."""
        x = np.asarray(speech_array, dtype=np.float32)
        return ((x - x.mean()) / np.sqrt(x.var() + 1e-7)).astype(np.float32)

    @torch.no_grad()
    def preprocess_array(self, speech_array, clip_length=-1, to_cpu=True):
        """audio_processor.py:105-129 from `speech_array, sampling_rate = librosa.load(...)` on."""
        feat = self.normalize(speech_array)
        seq_len = math.ceil(len(feat) / self.sample_rate * self.fps)
        audio_length = seq_len
        x = torch.from_numpy(feat).to(self.device)
        if clip_length > 0 and seq_len % clip_length != 0:
            pad = (clip_length - seq_len % clip_length) * (self.sample_rate // self.fps)
            x = torch.nn.functional.pad(x, (0, pad), "constant", 0.0)
            seq_len += clip_length - seq_len % clip_length
        out = self.audio_encoder(x.unsqueeze(0), seq_len=seq_len, output_hidden_states=True)
        if self.only_last_features:
            emb = out.last_hidden_state.squeeze(0)
        else:
            emb = torch.stack(out.hidden_states[1:], dim=1).squeeze(0).permute(1, 0, 2)      # "b s d -> s b d"
        emb = emb.float()
        return (emb.cpu() if to_cpu else emb.contiguous()), audio_length

    def preprocess(self, wav_file, clip_length=-1):
        return self.preprocess_array(load_wav(wav_file, self.sample_rate), clip_length)

    def get_embedding(self, wav_file):
        """audio_processor.py:130-164: `preprocess` without the clip_length padding; returns the embedding only."""
        return self.preprocess_array(load_wav(wav_file, self.sample_rate), clip_length=-1)[0]

    # context-manager surface of the reference (audio_processor.py:166-176)
    def close(self):
        return self

    def __enter__(self):
        return self

    def __exit__(self, _exc_type, _exc_val, _exc_tb):
        self.close()
