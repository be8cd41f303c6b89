"""The three worked examples of RFC 9639, Appendix D, as complete FLAC files.

These are the only vectors in this repository that were not produced by its own code.  Each one is self-verifying: the
CRC-8 of every frame header, the CRC-16 of every frame and the MD5 of the decoded samples stored in STREAMINFO have to
agree with the bytes below (tests/test_oracle.py checks all three before it trusts a vector), so a transcription error
cannot pass.  Between them they exercise: VERBATIM subframes with wasted bits (1), a SEEKTABLE / VORBIS_COMMENT /
PADDING prologue, side/right stereo with a 17-bit side channel, FIXED order-1 subframes, Rice partitions, a short last
frame (2), and a mono 8-bit LPC subframe (order 3, 4-bit coefficients, shift 2) with partition order 2 and an ESCAPED
partition (3).
"""

EXAMPLE_1 = bytes.fromhex(
    "664c6143" "80000022" "1000" "1000" "00000f" "00000f" "0ac442f0" "00000001" "3e84b41807dc690307586a3dad1a2e0f"
    "fff869180000bf" "0358fd" "03128b" "aa9a"
)

EXAMPLE_2 = bytes.fromhex(
    "664c6143" "00000022" "00100010" "000017" "000044" "0ac442f0" "00000013" "d5b0564975e98b8d8b930422757b8103"
    "03000012" "0000000000000000" "0000000000000000" "0010"
    "0400003a" "20000000" "7265666572656e6365206c6962464c414320312e332e33203230313930383034" "01000000" "0e000000"
    "5449544c453dd7a9d79cd795d79d"
    "81000006" "000000000000"
    "fff86998000f9912" "086701623d1442998f5df70d" "6fe00c17caeb21000ee7a77a" "24a1590c1217b603097b784f"
    "aa9a33d285e070ad5b1b4851" "b4010d99d2cd1a68f1e6b810"
    "fff869180102a402c382c40b" "c14a03ee48dd03b67c1330"
)

EXAMPLE_3 = bytes.fromhex(
    "664c6143" "80000022" "10001000" "00001f" "00001f" "07d00070" "00000018" "f8f9e396f5cbcfc6dc807f9977906b32"
    "fff868020017e9" "44004f6f313d1047d227cb6d09083145" "2bdc2822228057a3"
)

# (file, channels, bits per sample, samples per channel, byte offsets of the frames)
EXAMPLES = {
    "example1": (EXAMPLE_1, 2, 16, 1, (42,)),
    "example2": (EXAMPLE_2, 2, 16, 19, (0x88, 0xCC)),
    "example3": (EXAMPLE_3, 1, 8, 24, (42,)),
}
