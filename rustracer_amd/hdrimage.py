"""Host mirror of src/hdrimage.rs: HdrImage pixel store, PFM wire format, tone mapping, LDR output.

Output sink of the render core (HdrImage::set_pixels, src/hdrimage.rs:86-96) and texture source
for ImagePigment.  Pixel order: row 0 = top, (h, w, 3) float32.
"""
import io
import math

import numpy as np

F = np.float32
DELTA = F(1e-10)  # src/hdrimage.rs:17


class HdrImageErr(Exception):
    pass


class HdrImage:
    """src/hdrimage.rs:21-96."""

    def __init__(self, width, height):
        self.width, self.height = int(width), int(height)
        self.pixels = np.zeros((self.height, self.width, 3), dtype=F)

    def shape(self):
        return (self.width, self.height)

    def valid_coordinates(self, x, y):
        return 0 <= x < self.width and 0 <= y < self.height

    def get_pixel(self, x, y):
        if not self.valid_coordinates(x, y):
            raise HdrImageErr(f"pixel ({x}, {y}) out of bounds ({self.width}, {self.height})")
        return tuple(self.pixels[y, x])

    def set_pixel(self, x, y, color):
        if not self.valid_coordinates(x, y):
            raise HdrImageErr(f"pixel ({x}, {y}) out of bounds ({self.width}, {self.height})")
        self.pixels[y, x] = color

    def set_pixels(self, pixels):
        """src/hdrimage.rs:86-96: accepts exactly width*height colors."""
        px = np.asarray(pixels, dtype=F)
        if px.size != self.pixels.size:
            raise HdrImageErr(f"invalid pixels size {px.size // 3}, expected {self.pixels.size // 3}")
        self.pixels = px.reshape(self.height, self.width, 3).copy()

    # ---- PFM (src/hdrimage.rs:127-233) ----
    @staticmethod
    def read_pfm_image(stream):
        def line():
            raw = stream.readline()
            if not raw.endswith(b"\n"):
                raise HdrImageErr("invalid pfm file format: expected eol as separator inside header")
            try:
                return raw.decode("ascii").rstrip()
            except UnicodeDecodeError:
                raise HdrImageErr("impossible to read from pfm file: stream did not contain valid UTF-8")
        if line() != "PF":
            raise HdrImageErr("invalid pfm file format: wrong magic inside header")
        width, height = _parse_img_shape(line())
        endian = _parse_endianness(line())
        n = width * height * 3
        raw = stream.read(n * 4)
        if len(raw) != n * 4:
            raise HdrImageErr("impossible to read binary data from the file")
        if stream.read(1) != b"":
            raise HdrImageErr("invalid pfm file format: find binary content, expected eof")
        data = np.frombuffer(raw, dtype=("<f4" if endian == "little" else ">f4")).astype(F)
        img = HdrImage(width, height)
        img.pixels = data.reshape(height, width, 3)[::-1].copy()  # file stores bottom row first
        return img

    @staticmethod
    def read_pfm_file(path):
        with open(path, "rb") as f:
            return HdrImage.read_pfm_image(f)

    def write_pfm_image(self, stream, endianness="little"):
        stream.write(f"PF\n{self.width} {self.height}\n".encode("ascii"))
        stream.write(b"-1.0\n" if endianness == "little" else b"1.0\n")
        stream.write(self.pixels[::-1].astype("<f4" if endianness == "little" else ">f4").tobytes())

    def write_pfm_file(self, path, endianness="little"):
        with open(path, "wb") as f:
            self.write_pfm_image(f, endianness)

    # ---- tone mapping (src/hdrimage.rs:239-268) ----
    def average_luminosity(self):
        """Sequential f32 sum of f32::log10(DELTA + luminosity) in pixel order, then f32::powf(10, sum/N)
        (src/hdrimage.rs:239-245).  Color::luminosity reduces with f32::max / f32::min, which ignore a NaN operand
        (src/color.rs:33-36): fmax / fmin here."""
        px = self.pixels.reshape(-1, 3)
        lum = ((np.fmax.reduce(px, axis=1) + np.fmin.reduce(px, axis=1)) * F(0.5)).astype(F)
        with np.errstate(all="ignore"):
            logs = np.log10((DELTA + lum).astype(F))       # float32 in, float32 log10 out
        # order-sensitive f32 accumulation (hdrimage.rs:240-243): cumsum of float32 adds left to right in float32
        total = np.cumsum(logs, dtype=F)[-1] if len(logs) else F(0.0)
        with np.errstate(all="ignore"):
            return F(np.power(F(10.0), F(total / F(len(logs)))))

    def normalize_image(self, factor, luminosity=None):
        lum = self.average_luminosity() if luminosity is None else F(luminosity)
        self.pixels = (self.pixels * F(F(factor) / lum)).astype(F)

    def clamp_image(self):
        self.pixels = (self.pixels / (F(1.0) + self.pixels)).astype(F)

    # ---- LDR (src/hdrimage.rs:286-334) ----
    def ldr_bytes(self, gamma=1.0, depth=8):
        scale = 255.0 if depth == 8 else 65535.0
        g = F(F(1.0) / F(gamma))
        v = np.power(self.pixels.astype(np.float64), float(g)).astype(F)
        v = (F(scale) * v).astype(F)
        v = np.nan_to_num(v, nan=0.0, posinf=scale, neginf=0.0)
        v = np.clip(np.trunc(v), 0, scale)  # Rust `as u8` / `as u16`: truncate, saturate, NaN -> 0
        return v.astype(np.uint8 if depth == 8 else np.uint16)

    def write_ldr_file(self, path, gamma=1.0):
        path = str(path)
        ext = path.lower()                  # ImageFormat::from_path lower-cases the extension
        if ext.endswith(".png"):
            from PIL import Image
            Image.fromarray(self.ldr_bytes(gamma, 8), "RGB").save(path, format="PNG")
        elif ext.endswith(".ff"):
            rgb = self.ldr_bytes(gamma, 16)
            rgba = np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 65535, np.uint16)], axis=2)
            with open(path, "wb") as f:
                f.write(b"farbfeld" + self.width.to_bytes(4, "big") + self.height.to_bytes(4, "big"))
                f.write(rgba.astype(">u2").tobytes())
        else:
            raise HdrImageErr(f"unsupported ldr file format {path.rsplit('.', 1)[-1]!r}")


def _parse_img_shape(line):
    """src/hdrimage.rs:357-370: two u32 separated by one or more spaces."""
    parts = [p for p in line.split(" ") if p != ""]
    if len(parts) != 2:
        raise HdrImageErr("invalid pfm file format: wrong image shape inside header")
    dims = []
    for text, what in zip(parts, ("image width", "image height")):
        if not text.isdigit():  # u32::from_str rejects signs other than '+', decimals, letters
            if not (text.startswith("+") and text[1:].isdigit()):
                raise HdrImageErr(f"impossible to parse {what} from pfm header")
        v = int(text)
        if v > 0xFFFFFFFF:
            raise HdrImageErr(f"impossible to parse {what} from pfm header")
        dims.append(v)
    return dims[0], dims[1]


def _parse_endianness(line):
    try:
        v = float(line)
    except ValueError:
        raise HdrImageErr("impossible to parse endianness from pfm header")
    if v > 0:
        return "big"
    if v < 0:
        return "little"
    raise HdrImageErr("invalid pfm file format: invalid endianness inside header")
