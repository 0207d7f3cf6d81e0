"""ReferenceAttentionControl -- the ReferenceNet <-> denoising-UNet coupling.

Reference: hallo/models/mutual_self_attention.py:39-496.  There it monkey-patches the `forward` of
every (Temporal)BasicTransformerBlock (register_reference_hooks 115-402) and moves features through
`module.bank` lists.  Here the coupling is explicit: the writer UNet returns its 16 feature banks,
`update()` hands them to the reader UNet as fp16-rounded tensors (the reference hard-codes
`dtype=torch.float16`, :404,452-453 -- SURVEY F4), `clear()` drops them.  Constructor and method
signatures are the reference's so face_animate.py:300-313,395,429-430 read the same.
"""
import torch


class ReferenceAttentionControl:
    def __init__(self, unet, mode="write", do_classifier_free_guidance=False, attention_auto_machine_weight=float("inf"),
                 gn_auto_machine_weight=1.0, style_fidelity=1.0, reference_attn=True, reference_adain=False,
                 fusion_blocks="midup", batch_size=1):
        if mode not in ("read", "write"):
            raise AssertionError("mode must be 'read' or 'write'")
        if fusion_blocks not in ("midup", "full"):
            raise AssertionError("fusion_blocks must be 'midup' or 'full'")
        if fusion_blocks != "full":
            raise NotImplementedError("the Hallo inference path uses fusion_blocks='full' (face_animate.py:300-313)")
        self.unet, self.mode = unet, mode
        self.do_classifier_free_guidance = do_classifier_free_guidance
        self.fusion_blocks, self.batch_size = fusion_blocks, batch_size
        if mode == "read":
            unet.reference_do_cfg = bool(do_classifier_free_guidance)

    def update(self, writer, dtype=torch.float16):
        """Copy the writer's banks to the reader.  Pairing: i-th writer block <-> i-th reader block in
        module order (the reference's stable sort by -C keeps both sequences in the same relative order,
        :445-453)."""
        if self.mode != "read":
            raise AssertionError("update() is called on the reader")
        banks = writer.unet.written_banks
        if len(banks) == 0:
            raise RuntimeError("the writer UNet has not been run")
        self.unet.reference_bank = [b.clone().to(dtype) for b in banks]

    def clear(self):
        if self.mode == "read":
            self.unet.reference_bank = None
        else:
            self.unet.written_banks = []
